/* tables.c -- host-side construction of the immutable DSP tables (product code).
 *
 * Bit-exactness with the reference decoder depends on these being computed
 * the way the reference computes them: in double with libm, rounded to float
 * at the same point.  Each builder names the reference lines it matches.
 * C99; compiled without -ffast-math and with -ffp-contract=off.
 */
#include <math.h>
#include <string.h>
#include "tables.h"
#include "heaac_iso_tables.h"

static const double PI = 3.14159265358979323846;
static const double SQRT2 = 1.41421356237309504880;
static const double SQRT1_2 = 0.70710678118654752440;

/* ff_cos_N[i] = (float)cos(i * 2pi/N), i = 0..N/4          fft.c:67-79 */
static void fill_cos(float *dst, int n)
{
    const double step = 2 * PI / n;
    for (int i = 0; i <= n / 4; i++)
        dst[i] = (float)cos(i * step);
}

/* tcos/tsin of ff_mdct_init(nbits, 1, scale)               mdct.c:94-100 */
static void fill_rotation(float *dst, int n, double scale)
{
    const int q = n / 4;
    const double theta = 1.0 / 8.0 + (scale < 0 ? q : 0);
    const double amp = sqrt(fabs(scale));
    for (int i = 0; i < q; i++) {
        const double a = 2 * PI * (i + theta) / n;
        dst[i]     = (float)(-cos(a) * amp);
        dst[q + i] = (float)(-sin(a) * amp);
    }
}

/* Split-radix input order                                  fft.c:56-65,121-122 */
static int sr_index(int i, int n)
{
    if (n <= 2)
        return i & 1;
    int half = n >> 1, quarter = n >> 2;
    if (!(i & half))
        return sr_index(i, half) * 2;
    /* inverse transform: odd branch +1 when bit (n/4) is clear */
    return sr_index(i, quarter) * 4 + ((i & quarter) ? -1 : 1);
}

static void fill_revtab(uint16_t *dst, int n)
{
    for (int i = 0; i < n; i++)
        dst[(-sr_index(i, n)) & (n - 1)] = (uint16_t)i;
}

/* Kaiser-Bessel derived window                             mdct.c:35-54 */
static void fill_kbd(float *dst, double alpha_f, int n)
{
    /* `alpha` is a float argument in the reference; 4.0 and 6.0 are exact */
    const float alpha = (float)alpha_f;
    const double a2 = (alpha * PI / n) * (alpha * PI / n);
    double acc = 0.0;
    double cum[1024];
    for (int i = 0; i < n; i++) {
        const double x = i * (n - i) * a2;
        double bessel = 1.0;
        for (int j = 50; j > 0; j--)
            bessel = bessel * x / (j * j) + 1;
        acc += bessel;
        cum[i] = acc;
    }
    acc++;
    for (int i = 0; i < n; i++)
        dst[i] = (float)sqrt(cum[i] / acc);
}

/* Sine window                                              mdct_tablegen.h:49-53 */
static void fill_sine(float *dst, int n)
{
    for (int i = 0; i < n; i++)
        dst[i] = sinf((float)((i + 0.5) * (PI / (2.0 * n))));
}

/* 640-tap QMF window from the 321 ISO taps                 aacsbr.c:117-123 */
static void fill_qmf(float *us, float *ds)
{
    for (int i = 0; i <= 320; i++)
        us[i] = heaac_iso_qmf_c[i];
    for (int i = 1; i < 320; i++)
        us[320 + i] = us[320 - i];
    us[384] = -us[384];
    us[512] = -us[512];
    for (int i = 0; i < 320; i++)
        ds[i] = us[2 * i];
}

/* Hybrid filterbank prototypes -> modulated complex filters
 *                                                          aacps_tablegen.h:48-78 */
static void fill_hybrid(float *dst, const float *proto, int bands)
{
    for (int q = 0; q < bands; q++)
        for (int n = 0; n < 7; n++) {
            const double th = 2 * PI * (q + 0.5) * (n - 6) / bands;
            dst[(q * 7 + n) * 2 + 0] = (float)(proto[n] *  cos(th));
            dst[(q * 7 + n) * 2 + 1] = (float)(proto[n] * -sin(th));
        }
}

/* IPD/OPD smoothing table                                  aacps_tablegen.h:123-139 */
static void fill_pd_smooth(float *re, float *im)
{
    const float s = (float)SQRT1_2;
    const float sn[8] = { 0,  s, 1,  s,  0, -s, -1, -s };
    const float cs[8] = { 1,  s, 0, -s, -1, -s,  0,  s };
    for (int a = 0; a < 8; a++)
        for (int b = 0; b < 8; b++)
            for (int c = 0; c < 8; c++) {
                const float r = 0.25f * cs[a] + 0.5f * cs[b] + cs[c];
                const float i = 0.25f * sn[a] + 0.5f * sn[b] + sn[c];
                const float mag = (float)(1 / sqrt(i * i + r * r));
                re[a * 64 + b * 8 + c] = r * mag;
                im[a * 64 + b * 8 + c] = i * mag;
            }
}

/* Mixing matrices HA (mode A) and HB (mode B)              aacps_tablegen.h:141-172 */
static void fill_mixing(float *HA, float *HB)
{
    /* linear IID, default then fine quantisation (ISO/IEC 14496-3 Table 8.25/8.26) */
    static const float iid_lin[46] = {
        0.05623413251903, 0.12589254117942, 0.19952623149689, 0.31622776601684,
        0.44668359215096, 0.63095734448019, 0.79432823472428, 1,
        1.25892541179417, 1.58489319246111, 2.23872113856834, 3.16227766016838,
        5.01187233627272, 7.94328234724282, 17.7827941003892,
        0.00316227766017, 0.00562341325190, 0.01,             0.01778279410039,
        0.03162277660168, 0.05623413251903, 0.07943282347243, 0.11220184543020,
        0.15848931924611, 0.22387211385683, 0.31622776601684, 0.39810717055350,
        0.50118723362727, 0.63095734448019, 0.79432823472428, 1,
        1.25892541179417, 1.58489319246111, 1.99526231496888, 2.51188643150958,
        3.16227766016838, 4.46683592150963, 6.30957344480193, 8.91250938133745,
        12.5892541179417, 17.7827941003892, 31.6227766016838, 56.2341325190349,
        100,              177.827941003892, 316.227766016837,
    };
    static const float icc[8]      = { 1, 0.937, 0.84118, 0.60092, 0.36764, 0, -0.589, -1 };
    static const float acos_icc[8] = { 0, 0.35685527, 0.57133466, 0.92614472, 1.1943263,
                                       3.14159265358979323846 / 2, 2.2006171,
                                       3.14159265358979323846 };
    for (int q = 0; q < 46; q++) {
        const float c  = iid_lin[q];
        const float c1 = (float)SQRT2 / sqrtf(1.0f + c * c);
        const float c2 = c * c1;
        for (int r = 0; r < 8; r++) {
            float *a = HA + (q * 8 + r) * 4, *b = HB + (q * 8 + r) * 4;
            {   /* mode A */
                const float alpha = 0.5f * acos_icc[r];
                const float beta  = alpha * (c1 - c2) * (float)SQRT1_2;
                a[0] = c2 * cosf(beta + alpha);
                a[1] = c1 * cosf(beta - alpha);
                a[2] = c2 * sinf(beta + alpha);
                a[3] = c1 * sinf(beta - alpha);
            }
            {   /* mode B */
                const float rho = icc[r] > 0.05f ? icc[r] : 0.05f;
                float alpha = 0.5f * atan2f(2.0f * c * rho, c * c - 1.0f);
                float mu = c + 1.0f / c;
                mu = sqrtf(1 + (4 * rho * rho - 4) / (mu * mu));
                const float gamma = atanf(sqrtf((1.0f - mu) / (1.0f + mu)));
                if (alpha < 0)
                    alpha = (float)(alpha + PI / 2);
                const float ac = cosf(alpha), as = sinf(alpha);
                const float gc = cosf(gamma), gs = sinf(gamma);
                b[0] = (float)( SQRT2 * ac * gc);
                b[1] = (float)( SQRT2 * as * gc);
                b[2] = (float)(-SQRT2 * as * gs);
                b[3] = (float)( SQRT2 * ac * gs);
            }
        }
    }
}

/* All-pass fractional delays and phi_fract                 aacps_tablegen.h:174-203 */
static void fill_allpass(float *Q, float *phi)
{
    static const signed char centre20[10] = { -3, -1, 1, 3, 5, 7, 10, 14, 18, 22 };
    static const signed char centre34[32] = {
         2,  6, 10, 14, 18, 22, 26, 30, 34,-10, -6, -2, 51, 57, 15, 21,
        27, 33, 39, 45, 54, 66, 78, 42,102, 66, 78, 90,102,114,126, 90,
    };
    static const float link_frac[3] = { 0.43f, 0.75f, 0.347f };
    const float gain_frac = 0.39f;
    for (int mode = 0; mode < 2; mode++) {
        const int nb = mode ? 50 : 30;
        for (int k = 0; k < nb; k++) {
            double fc;
            if (mode == 0)
                fc = k < 10 ? centre20[k] * 0.125 : (double)(k - 6.5f);
            else
                fc = k < 32 ? centre34[k] / 24. : (double)(k - 26.5f);
            for (int m = 0; m < 3; m++) {
                const double th = -PI * link_frac[m] * fc;
                Q[((mode * 50 + k) * 3 + m) * 2 + 0] = (float)cos(th);
                Q[((mode * 50 + k) * 3 + m) * 2 + 1] = (float)sin(th);
            }
            const double th = -PI * gain_frac * fc;
            phi[(mode * 50 + k) * 2 + 0] = (float)cos(th);
            phi[(mode * 50 + k) * 2 + 1] = (float)sin(th);
        }
    }
}

void heaac_build_tables(HeaacHostTables *t)
{
    /* ISO/IEC 14496-3 Table 8.A.x hybrid prototype filters (aacps_tablegen.h:48-66,
     * aacpsdata.c:160-163) */
    static const float g0_Q8[7]  = { 0.00746082949812f, 0.02270420949825f, 0.04546865930473f,
                                     0.07266113929591f, 0.09885108575264f, 0.11793710567217f, 0.125f };
    static const float g0_Q12[7] = { 0.04081179924692f, 0.03812810994926f, 0.05144908135699f,
                                     0.06399831151592f, 0.07428313801106f, 0.08100347892914f,
                                     0.08333333333333f };
    static const float g1_Q8[7]  = { 0.01565675600122f, 0.03752716391991f, 0.05417891378782f,
                                     0.08417044116767f, 0.10307344158036f, 0.12222452249753f, 0.125f };
    static const float g2_Q4[7]  = { -0.05908211155639f, -0.04871498374946f, 0.0f,
                                     0.07778723915851f, 0.16486303567403f, 0.23279856662996f, 0.25f };
    static const float g1_Q2[7]  = { 0.0f, 0.01899487526049f, 0.0f, -0.07293139167538f,
                                     0.0f, 0.30596630545168f, 0.5f };
    float *f = t->f;
    memset(t, 0, sizeof(*t));

    fill_cos(f + TB_COS16, 16);
    fill_cos(f + TB_COS32, 32);
    fill_cos(f + TB_COS64, 64);
    fill_cos(f + TB_COS128, 128);
    fill_cos(f + TB_COS256, 256);
    fill_cos(f + TB_COS512, 512);

    fill_rotation(f + TB_ROT2048, 2048, 1.0);        /* aacdec.c:590 */
    fill_rotation(f + TB_ROT256,   256, 1.0);        /* aacdec.c:591 */
    fill_rotation(f + TB_ROT128S,  128, 1.0 / 64);   /* aacsbr.c:134 */
    fill_rotation(f + TB_ROT128A,  128, -2.0);       /* aacsbr.c:135 */

    fill_revtab(t->rev + RV_512, 512);
    fill_revtab(t->rev + RV_64, 64);
    fill_revtab(t->rev + RV_32, 32);

    fill_kbd(f + TB_KBD_LONG, 4.0, 1024);            /* aacdec.c:593 */
    fill_kbd(f + TB_KBD_SHORT, 6.0, 128);            /* aacdec.c:594 */
    fill_sine(f + TB_SINE_LONG, 1024);
    fill_sine(f + TB_SINE_SHORT, 128);

    fill_qmf(f + TB_QMF_US, f + TB_QMF_DS);
    memcpy(f + TB_NOISE, heaac_iso_noise, 1024 * sizeof(float));

    fill_pd_smooth(f + TB_PD_RE, f + TB_PD_IM);
    fill_mixing(f + TB_HA, f + TB_HB);
    fill_hybrid(f + TB_F20_0_8,  g0_Q8,   8);
    fill_hybrid(f + TB_F34_0_12, g0_Q12, 12);
    fill_hybrid(f + TB_F34_1_8,  g1_Q8,   8);
    fill_hybrid(f + TB_F34_2_4,  g2_Q4,   4);
    fill_allpass(f + TB_QFRACT, f + TB_PHIFRACT);
    memcpy(f + TB_G1_Q2, g1_Q2, sizeof(g1_Q2));
    for (int k = 0; k < 512; k++) {
        const int e = t->rev[RV_512 + k], slot = (e & 15) * 32 + (e >> 4);
        f[TB_ROTA512 + 2 * slot]     = f[TB_ROT2048 + k];
        f[TB_ROTA512 + 2 * slot + 1] = f[TB_ROT2048 + 512 + k];
    }
    for (int k = 0; k < 64; k++) {
        const int e = t->rev[RV_64 + k], slot = (e & 15) * 4 + (e >> 4);
        f[TB_ROTA64 + 2 * slot]     = f[TB_ROT256 + k];
        f[TB_ROTA64 + 2 * slot + 1] = f[TB_ROT256 + 64 + k];
    }
}

int heaac_get_table(const char *name, float *dst, int max)
{
    static HeaacHostTables T;
    static int ready;
    static const struct { const char *nm; int off, cnt; } idx[] = {
        { "cos16", TB_COS16, 5 }, { "cos32", TB_COS32, 9 }, { "cos64", TB_COS64, 17 },
        { "cos128", TB_COS128, 33 }, { "cos256", TB_COS256, 65 }, { "cos512", TB_COS512, 129 },
        { "tcos2048", TB_ROT2048, 1024 }, { "tcos256", TB_ROT256, 128 },
        { "tcos128s", TB_ROT128S, 64 }, { "tcos128a", TB_ROT128A, 64 },
        { "kbd_long", TB_KBD_LONG, 1024 }, { "kbd_short", TB_KBD_SHORT, 128 },
        { "sine_long", TB_SINE_LONG, 1024 }, { "sine_short", TB_SINE_SHORT, 128 },
        { "qmf_us", TB_QMF_US, 640 }, { "qmf_ds", TB_QMF_DS, 320 }, { "noise", TB_NOISE, 1024 },
        { "pd_re_smooth", TB_PD_RE, 512 }, { "pd_im_smooth", TB_PD_IM, 512 },
        { "HA", TB_HA, 1472 }, { "HB", TB_HB, 1472 },
        { "f20_0_8", TB_F20_0_8, 112 }, { "f34_0_12", TB_F34_0_12, 168 },
        { "f34_1_8", TB_F34_1_8, 112 }, { "f34_2_4", TB_F34_2_4, 56 },
        { "Q_fract_allpass", TB_QFRACT, 600 }, { "phi_fract", TB_PHIFRACT, 200 },
    };
    if (!ready) {
        heaac_build_tables(&T);
        ready = 1;
    }
    if (!strncmp(name, "revtab", 6) && name[6] >= '0' && name[6] <= '3') {
        const int w = name[6] - '0';
        const int off = w == 0 ? RV_512 : w == 1 ? RV_64 : RV_32;
        const int n = w == 0 ? 512 : w == 1 ? 64 : 32;
        if (n > max)
            return -1;
        for (int i = 0; i < n; i++)
            dst[i] = T.rev[off + i];
        return n;
    }
    for (unsigned i = 0; i < sizeof(idx) / sizeof(idx[0]); i++)
        if (!strcmp(name, idx[i].nm)) {
            if (idx[i].cnt > max)
                return -1;
            memcpy(dst, T.f + idx[i].off, idx[i].cnt * sizeof(float));
            return idx[i].cnt;
        }
    return -1;
}
