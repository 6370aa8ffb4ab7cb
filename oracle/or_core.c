/* or_core.c -- oracle: tables, split-radix FFT, IMDCT, AAC-LC windowing,
 * float->int16.  TEST INFRASTRUCTURE (see oracle.h).
 *
 * All float arithmetic keeps the reference's expression shapes; compile with
 * -ffp-contract=off.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "heaac_iso_tables.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#ifndef M_SQRT1_2
#define M_SQRT1_2 0.70710678118654752440
#endif
#ifndef M_SQRT2
#define M_SQRT2 1.41421356237309504880
#endif

static or_tables T;

/* ------------------------------------------------------------------ */
/* tables                                                              */
/* ------------------------------------------------------------------ */

/* fft.c:56-65 split_radix_permutation */
static int sr_perm(int i, int n, int inverse)
{
    int m;
    if (n <= 2)
        return i & 1;
    m = n >> 1;
    if (!(i & m))
        return sr_perm(i, m, inverse) * 2;
    m >>= 1;
    if (inverse == !(i & m))
        return sr_perm(i, m, inverse) * 4 + 1;
    return sr_perm(i, m, inverse) * 4 - 1;
}

/* fft.c:67-79 ff_init_ff_cos_tabs: cos(2*pi*i/m) for i <= m/4, mirrored */
static void build_cos_tab(int bits)
{
    int m = 1 << bits, i;
    double freq = 2 * M_PI / m;
    float *tab = malloc(sizeof(float) * (m / 2 + 1));
    for (i = 0; i <= m / 4; i++)
        tab[i] = cos(i * freq);
    for (i = 1; i < m / 4; i++)
        tab[m / 2 - i] = tab[i];
    T.cos_tab[bits] = tab;
}

/* mdct.c:61-105 ff_mdct_init (inverse = 1, FF_MDCT_PERM_NONE) + fft.c:121-122 */
static void build_mdct(or_mdct *s, int nbits, double scale)
{
    int n = 1 << nbits, n4 = n >> 2, i;
    double theta, alpha;
    s->nbits = nbits;
    s->n = n;
    s->revtab = malloc(sizeof(uint16_t) * n4);
    for (i = 0; i < n4; i++)
        s->revtab[-sr_perm(i, n4, 1) & (n4 - 1)] = i;
    s->tcos = malloc(sizeof(float) * n4 * 2);
    s->tsin = s->tcos + n4;
    theta = 1.0 / 8.0 + (scale < 0 ? n4 : 0);
    scale = sqrt(fabs(scale));
    for (i = 0; i < n4; i++) {
        alpha = 2 * M_PI * (i + theta) / n;
        s->tcos[i] = -cos(alpha) * scale;
        s->tsin[i] = -sin(alpha) * scale;
    }
}

/* mdct.c:35-54 ff_kbd_window_init */
static void build_kbd(float *window, float alpha, int n)
{
    int i, j;
    double sum = 0.0, bessel, tmp;
    double *local = malloc(sizeof(double) * n);
    double alpha2 = (alpha * M_PI / n) * (alpha * M_PI / n);
    for (i = 0; i < n; i++) {
        tmp = i * (n - i) * alpha2;
        bessel = 1.0;
        for (j = 50; j > 0; j--)
            bessel = bessel * tmp / (j * j) + 1;
        sum += bessel;
        local[i] = sum;
    }
    sum++;
    for (i = 0; i < n; i++)
        window[i] = sqrt(local[i] / sum);
    free(local);
}

/* mdct_tablegen.h:49-53 ff_sine_window_init */
static void build_sine(float *window, int n)
{
    int i;
    for (i = 0; i < n; i++)
        window[i] = sinf((i + 0.5) * (M_PI / (2.0 * n)));
}

/* aacps_tablegen.h:68-78 */
static void ps_filters(float (*filter)[7][2], const float *proto, int bands)
{
    int q, n;
    for (q = 0; q < bands; q++) {
        for (n = 0; n < 7; n++) {
            double theta = 2 * M_PI * (q + 0.5) * (n - 6) / bands;
            filter[q][n][0] = proto[n] *  cos(theta);
            filter[q][n][1] = proto[n] * -sin(theta);
        }
    }
}

/* aacps_tablegen.h:80-209 ps_tableinit */
static void build_ps_tables(void)
{
    static const float ipdopd_sin[] = { 0, M_SQRT1_2, 1,  M_SQRT1_2,  0, -M_SQRT1_2, -1, -M_SQRT1_2 };
    static const float ipdopd_cos[] = { 1, M_SQRT1_2, 0, -M_SQRT1_2, -1, -M_SQRT1_2,  0,  M_SQRT1_2 };
    static const float iid_par_dequant[] = {
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
    static const float icc_invq[] = { 1, 0.937, 0.84118, 0.60092, 0.36764, 0, -0.589, -1 };
    static const float acos_icc_invq[] = {
        0, 0.35685527, 0.57133466, 0.92614472, 1.1943263, M_PI / 2, 2.2006171, M_PI
    };
    static const int8_t f_center_20[] = { -3, -1, 1, 3, 5, 7, 10, 14, 18, 22 };
    static const int8_t f_center_34[] = {
         2,  6, 10, 14, 18, 22, 26, 30, 34,-10, -6, -2, 51, 57, 15, 21,
        27, 33, 39, 45, 54, 66, 78, 42,102, 66, 78, 90,102,114,126, 90,
    };
    static const float fractional_delay_links[] = { 0.43f, 0.75f, 0.347f };
    const float fractional_delay_gain = 0.39f;
    static const float g0_Q8[] = {
        0.00746082949812f, 0.02270420949825f, 0.04546865930473f, 0.07266113929591f,
        0.09885108575264f, 0.11793710567217f, 0.125f };
    static const float g0_Q12[] = {
        0.04081179924692f, 0.03812810994926f, 0.05144908135699f, 0.06399831151592f,
        0.07428313801106f, 0.08100347892914f, 0.08333333333333f };
    static const float g1_Q8[] = {
        0.01565675600122f, 0.03752716391991f, 0.05417891378782f, 0.08417044116767f,
        0.10307344158036f, 0.12222452249753f, 0.125f };
    static const float g2_Q4[] = {
        -0.05908211155639f, -0.04871498374946f, 0.0f, 0.07778723915851f,
         0.16486303567403f,  0.23279856662996f, 0.25f };
    int pd0, pd1, pd2, iid, icc, k, m;

    for (pd0 = 0; pd0 < 8; pd0++) {
        float pd0_re = ipdopd_cos[pd0], pd0_im = ipdopd_sin[pd0];
        for (pd1 = 0; pd1 < 8; pd1++) {
            float pd1_re = ipdopd_cos[pd1], pd1_im = ipdopd_sin[pd1];
            for (pd2 = 0; pd2 < 8; pd2++) {
                float pd2_re = ipdopd_cos[pd2], pd2_im = ipdopd_sin[pd2];
                float re_smooth = 0.25f * pd0_re + 0.5f * pd1_re + pd2_re;
                float im_smooth = 0.25f * pd0_im + 0.5f * pd1_im + pd2_im;
                float pd_mag = 1 / sqrt(im_smooth * im_smooth + re_smooth * re_smooth);
                T.pd_re_smooth[pd0 * 64 + pd1 * 8 + pd2] = re_smooth * pd_mag;
                T.pd_im_smooth[pd0 * 64 + pd1 * 8 + pd2] = im_smooth * pd_mag;
            }
        }
    }
    for (iid = 0; iid < 46; iid++) {
        float c = iid_par_dequant[iid];
        float c1 = (float)M_SQRT2 / sqrtf(1.0f + c * c);
        float c2 = c * c1;
        for (icc = 0; icc < 8; icc++) {
            {
                float alpha = 0.5f * acos_icc_invq[icc];
                float beta  = alpha * (c1 - c2) * (float)M_SQRT1_2;
                T.HA[iid][icc][0] = c2 * cosf(beta + alpha);
                T.HA[iid][icc][1] = c1 * cosf(beta - alpha);
                T.HA[iid][icc][2] = c2 * sinf(beta + alpha);
                T.HA[iid][icc][3] = c1 * sinf(beta - alpha);
            }
            {
                float alpha, gamma, mu, rho;
                float alpha_c, alpha_s, gamma_c, gamma_s;
                rho = icc_invq[icc] > 0.05f ? icc_invq[icc] : 0.05f;
                alpha = 0.5f * atan2f(2.0f * c * rho, c * c - 1.0f);
                mu = c + 1.0f / c;
                mu = sqrtf(1 + (4 * rho * rho - 4) / (mu * mu));
                gamma = atanf(sqrtf((1.0f - mu) / (1.0f + mu)));
                if (alpha < 0) alpha += M_PI / 2;
                alpha_c = cosf(alpha);
                alpha_s = sinf(alpha);
                gamma_c = cosf(gamma);
                gamma_s = sinf(gamma);
                T.HB[iid][icc][0] =  M_SQRT2 * alpha_c * gamma_c;
                T.HB[iid][icc][1] =  M_SQRT2 * alpha_s * gamma_c;
                T.HB[iid][icc][2] = -M_SQRT2 * alpha_s * gamma_s;
                T.HB[iid][icc][3] =  M_SQRT2 * alpha_c * gamma_s;
            }
        }
    }
    for (k = 0; k < 30; k++) {
        double f_center, theta;
        if (k < 10)
            f_center = f_center_20[k] * 0.125;
        else
            f_center = k - 6.5f;
        for (m = 0; m < 3; m++) {
            theta = -M_PI * fractional_delay_links[m] * f_center;
            T.Q_fract_allpass[0][k][m][0] = cos(theta);
            T.Q_fract_allpass[0][k][m][1] = sin(theta);
        }
        theta = -M_PI * fractional_delay_gain * f_center;
        T.phi_fract[0][k][0] = cos(theta);
        T.phi_fract[0][k][1] = sin(theta);
    }
    for (k = 0; k < 50; k++) {
        double f_center, theta;
        if (k < 32)
            f_center = f_center_34[k] / 24.;
        else
            f_center = k - 26.5f;
        for (m = 0; m < 3; m++) {
            theta = -M_PI * fractional_delay_links[m] * f_center;
            T.Q_fract_allpass[1][k][m][0] = cos(theta);
            T.Q_fract_allpass[1][k][m][1] = sin(theta);
        }
        theta = -M_PI * fractional_delay_gain * f_center;
        T.phi_fract[1][k][0] = cos(theta);
        T.phi_fract[1][k][1] = sin(theta);
    }
    ps_filters(T.f20_0_8,  g0_Q8,   8);
    ps_filters(T.f34_0_12, g0_Q12, 12);
    ps_filters(T.f34_1_8,  g1_Q8,   8);
    ps_filters(T.f34_2_4,  g2_Q4,   4);
}

const or_tables *oracle_tables(void)
{
    int b, n;
    if (T.ready)
        return &T;
    for (b = 4; b <= 9; b++)
        build_cos_tab(b);
    build_mdct(&T.mdct[0], 11, 1.0);        /* aacdec.c:590 */
    build_mdct(&T.mdct[1],  8, 1.0);        /* aacdec.c:591 */
    build_mdct(&T.mdct[2],  7, 1.0 / 64);   /* aacsbr.c:134 */
    build_mdct(&T.mdct[3],  7, -2.0);       /* aacsbr.c:135 */
    build_kbd(T.kbd_long, 4.0, 1024);       /* aacdec.c:593 */
    build_kbd(T.kbd_short, 6.0, 128);       /* aacdec.c:594 */
    build_sine(T.sine_long, 1024);
    build_sine(T.sine_short, 128);
    /* aacsbr.c:117-123 */
    for (n = 0; n <= 320; n++)
        T.qmf_us[n] = heaac_iso_qmf_c[n];
    for (n = 1; n < 320; n++)
        T.qmf_us[320 + n] = T.qmf_us[320 - n];
    T.qmf_us[384] = -T.qmf_us[384];
    T.qmf_us[512] = -T.qmf_us[512];
    for (n = 0; n < 320; n++)
        T.qmf_ds[n] = T.qmf_us[2 * n];
    memcpy(T.noise, heaac_iso_noise, sizeof(T.noise));
    build_ps_tables();
    T.ready = 1;
    return &T;
}

int oracle_get_table(const char *name, float *dst, int max)
{
    const or_tables *t = oracle_tables();
    const float *src = NULL;
    int n = 0, i;
#define TAB(nm, ptr, cnt) if (!strcmp(name, nm)) { src = (const float *)(ptr); n = (cnt); }
    TAB("cos16", t->cos_tab[4], 9)   TAB("cos32", t->cos_tab[5], 17)
    TAB("cos64", t->cos_tab[6], 33)  TAB("cos128", t->cos_tab[7], 65)
    TAB("cos256", t->cos_tab[8], 129) TAB("cos512", t->cos_tab[9], 257)
    TAB("tcos2048", t->mdct[0].tcos, 1024) TAB("tcos256", t->mdct[1].tcos, 128)
    TAB("tcos128s", t->mdct[2].tcos, 64)   TAB("tcos128a", t->mdct[3].tcos, 64)
    TAB("kbd_long", t->kbd_long, 1024) TAB("kbd_short", t->kbd_short, 128)
    TAB("sine_long", t->sine_long, 1024) TAB("sine_short", t->sine_short, 128)
    TAB("qmf_us", t->qmf_us, 640) TAB("qmf_ds", t->qmf_ds, 320)
    TAB("noise", t->noise, 1024)
    TAB("pd_re_smooth", t->pd_re_smooth, 512) TAB("pd_im_smooth", t->pd_im_smooth, 512)
    TAB("HA", t->HA, 46 * 8 * 4) TAB("HB", t->HB, 46 * 8 * 4)
    TAB("f20_0_8", t->f20_0_8, 8 * 14) TAB("f34_0_12", t->f34_0_12, 12 * 14)
    TAB("f34_1_8", t->f34_1_8, 8 * 14) TAB("f34_2_4", t->f34_2_4, 4 * 14)
    TAB("Q_fract_allpass", t->Q_fract_allpass, 2 * 50 * 3 * 2)
    TAB("phi_fract", t->phi_fract, 2 * 50 * 2)
#undef TAB
    if (!strncmp(name, "revtab", 6)) {
        int w = name[6] - '0';
        if (w < 0 || w > 3) return -1;
        n = t->mdct[w].n / 4;
        if (n > max) return -1;
        for (i = 0; i < n; i++) dst[i] = t->mdct[w].revtab[i];
        return n;
    }
    if (!src || n > max)
        return -1;
    memcpy(dst, src, sizeof(float) * n);
    return n;
}

/* ------------------------------------------------------------------ */
/* split-radix FFT, fft.c:213-367                                       */
/* ------------------------------------------------------------------ */

/* fft.c:292-304 */
static void sr_fft4(or_cpx *z)
{
    float t1, t2, t3, t4, t5, t6, t7, t8;
    t3 = z[0].re - z[1].re;  t1 = z[0].re + z[1].re;
    t8 = z[3].re - z[2].re;  t6 = z[3].re + z[2].re;
    z[2].re = t1 - t6;       z[0].re = t1 + t6;
    t4 = z[0].im - z[1].im;  t2 = z[0].im + z[1].im;
    t7 = z[2].im - z[3].im;  t5 = z[2].im + z[3].im;
    z[3].im = t4 - t8;       z[1].im = t4 + t8;
    z[3].re = t3 - t7;       z[1].re = t3 + t7;
    z[2].im = t2 - t5;       z[0].im = t2 + t5;
}

/* one TRANSFORM / TRANSFORM_ZERO + BUTTERFLIES, fft.c:213-254 */
static void sr_transform(or_cpx *a0, or_cpx *a1, or_cpx *a2, or_cpx *a3,
                         float wre, float wim, int zero)
{
    float t1, t2, t3, t4, t5, t6;
    if (zero) {
        t1 = a2->re; t2 = a2->im; t5 = a3->re; t6 = a3->im;
    } else {
        t1 = a2->re * wre + a2->im * wim;
        t2 = a2->im * wre - a2->re * wim;
        t5 = a3->re * wre - a3->im * wim;
        t6 = a3->im * wre + a3->re * wim;
    }
    t3 = t5 - t1;            t5 = t5 + t1;
    a2->re = a0->re - t5;    a0->re = a0->re + t5;
    a3->im = a1->im - t3;    a1->im = a1->im + t3;
    t4 = t2 - t6;            t6 = t2 + t6;
    a3->re = a1->re - t4;    a1->re = a1->re + t4;
    a2->im = a0->im - t6;    a0->im = a0->im + t6;
}

/* fft.c:306-324 */
static void sr_fft8(or_cpx *z)
{
    float t1, t2, t3, t4, t7, t8;
    sr_fft4(z);
    /* BF(t1, z[5].re, z[4].re, -z[5].re) etc: x - (-y) == x + y exactly */
    t1 = z[4].re + z[5].re;  z[5].re = z[4].re - z[5].re;
    t2 = z[4].im + z[5].im;  z[5].im = z[4].im - z[5].im;
    t3 = z[6].re + z[7].re;  z[7].re = z[6].re - z[7].re;
    t4 = z[6].im + z[7].im;  z[7].im = z[6].im - z[7].im;
    t8 = t3 - t1;            t1 = t3 + t1;
    t7 = t2 - t4;            t2 = t2 + t4;
    z[4].re = z[0].re - t1;  z[0].re = z[0].re + t1;
    z[4].im = z[0].im - t2;  z[0].im = z[0].im + t2;
    z[6].re = z[2].re - t7;  z[2].re = z[2].re + t7;
    z[6].im = z[2].im - t8;  z[2].im = z[2].im + t8;
    sr_transform(&z[1], &z[3], &z[5], &z[7], (float)M_SQRT1_2, (float)M_SQRT1_2, 0);
}

/* fft.c:257-281 pass(z, ff_cos_N, N/8): element k uses (cos[k], cos[N/4-k]).
 * fft16 (fft.c:327-339) is the same arithmetic with ff_cos_16[2] == sqrthalf. */
static void sr_pass(or_cpx *z, const float *cs, int n)
{
    int q = n >> 2, k;
    for (k = 0; k < q; k++)
        sr_transform(&z[k], &z[k + q], &z[k + 2 * q], &z[k + 3 * q],
                     cs[k], cs[q - k], k == 0);
}

/* fft.c:283-290 DECL_FFT */
static void sr_fft(or_cpx *z, int n, int bits)
{
    if (n == 4) { sr_fft4(z); return; }
    if (n == 8) { sr_fft8(z); return; }
    sr_fft(z, n / 2, bits - 1);
    sr_fft(z + n / 2, n / 4, bits - 2);
    sr_fft(z + 3 * (n / 4), n / 4, bits - 2);
    sr_pass(z, T.cos_tab[bits], n);
}

void oracle_fft_calc(int nbits, or_cpx *z)
{
    oracle_tables();
    sr_fft(z, 1 << nbits, nbits);
}

/* ------------------------------------------------------------------ */
/* IMDCT, mdct.c:107-179                                                */
/* ------------------------------------------------------------------ */

void oracle_imdct_half(int which, float *out, const float *in)
{
    const or_mdct *s = &oracle_tables()->mdct[which];
    int n = s->n, n2 = n >> 1, n4 = n >> 2, n8 = n >> 3, k;
    or_cpx *z = (or_cpx *)out;
    /* pre-rotation, mdct.c:139-146: CMUL(z[j], (in2, in1), (tcos, tsin)) */
    for (k = 0; k < n4; k++) {
        int j = s->revtab[k];
        float are = in[n2 - 1 - 2 * k], aim = in[2 * k];
        float bre = s->tcos[k], bim = s->tsin[k];
        z[j].re = are * bre - aim * bim;
        z[j].im = are * bim + aim * bre;
    }
    sr_fft(z, n4, s->nbits - 2);
    /* post-rotation + reordering, mdct.c:149-158 */
    for (k = 0; k < n8; k++) {
        float r0, i0, r1, i1;
        {
            float are = z[n8 - k - 1].im, aim = z[n8 - k - 1].re;
            float bre = s->tsin[n8 - k - 1], bim = s->tcos[n8 - k - 1];
            r0 = are * bre - aim * bim;
            i1 = are * bim + aim * bre;
        }
        {
            float are = z[n8 + k].im, aim = z[n8 + k].re;
            float bre = s->tsin[n8 + k], bim = s->tcos[n8 + k];
            r1 = are * bre - aim * bim;
            i0 = are * bim + aim * bre;
        }
        z[n8 - k - 1].re = r0;
        z[n8 - k - 1].im = i0;
        z[n8 + k].re = r1;
        z[n8 + k].im = i1;
    }
}

void oracle_imdct_calc(int which, float *out, const float *in)
{
    const or_mdct *s = &oracle_tables()->mdct[which];
    int n = s->n, n2 = n >> 1, n4 = n >> 2, k;
    oracle_imdct_half(which, out + n4, in);
    for (k = 0; k < n4; k++) {
        out[k] = -out[n2 - k - 1];
        out[n - k - 1] = out[n2 + k];
    }
}

/* ------------------------------------------------------------------ */
/* AAC-LC windowing + overlap-add                                       */
/* ------------------------------------------------------------------ */

/* dsputil.c:3832-3845 ff_vector_fmul_window_c */
static void fmul_window(float *dst, const float *src0, const float *src1,
                        const float *win, float add_bias, int len)
{
    int i, j;
    dst += len; win += len; src0 += len;
    for (i = -len, j = len - 1; i < 0; i++, j--) {
        float s0 = src0[i], s1 = src1[j], wi = win[i], wj = win[j];
        dst[i] = s0 * wj - s1 * wi + add_bias;
        dst[j] = s0 * wi + s1 * wj + add_bias;
    }
}

/* aacdec.c:1741-1806 */
void oracle_imdct_and_windowing(const float *in, const HeaacIcs *ics,
                                float *saved, float *out, float bias)
{
    const or_tables *t = oracle_tables();
    const float *swindow      = ics->use_kb_window[0] ? t->kbd_short : t->sine_short;
    const float *lwindow_prev = ics->use_kb_window[1] ? t->kbd_long  : t->sine_long;
    const float *swindow_prev = ics->use_kb_window[1] ? t->kbd_short : t->sine_short;
    const int ws0 = ics->window_sequence[0], ws1 = ics->window_sequence[1];
    float buf[1024], temp[128];
    int i;

    if (ws0 == HEAAC_EIGHT_SHORT_SEQUENCE) {
        for (i = 0; i < 1024; i += 128)
            oracle_imdct_half(1, buf + i, in + i);
    } else
        oracle_imdct_half(0, buf, in);

    if ((ws1 == HEAAC_ONLY_LONG_SEQUENCE || ws1 == HEAAC_LONG_STOP_SEQUENCE) &&
        (ws0 == HEAAC_ONLY_LONG_SEQUENCE || ws0 == HEAAC_LONG_START_SEQUENCE)) {
        fmul_window(out, saved, buf, lwindow_prev, bias, 512);
    } else {
        for (i = 0; i < 448; i++)
            out[i] = saved[i] + bias;
        if (ws0 == HEAAC_EIGHT_SHORT_SEQUENCE) {
            fmul_window(out + 448 + 0 * 128, saved + 448,        buf + 0 * 128, swindow_prev, bias, 64);
            fmul_window(out + 448 + 1 * 128, buf + 0 * 128 + 64, buf + 1 * 128, swindow,      bias, 64);
            fmul_window(out + 448 + 2 * 128, buf + 1 * 128 + 64, buf + 2 * 128, swindow,      bias, 64);
            fmul_window(out + 448 + 3 * 128, buf + 2 * 128 + 64, buf + 3 * 128, swindow,      bias, 64);
            fmul_window(temp,                buf + 3 * 128 + 64, buf + 4 * 128, swindow,      bias, 64);
            memcpy(out + 448 + 4 * 128, temp, 64 * sizeof(float));
        } else {
            fmul_window(out + 448, saved + 448, buf, swindow_prev, bias, 64);
            for (i = 576; i < 1024; i++)
                out[i] = buf[i - 512] + bias;
        }
    }

    if (ws0 == HEAAC_EIGHT_SHORT_SEQUENCE) {
        for (i = 0; i < 64; i++)
            saved[i] = temp[64 + i] - bias;
        fmul_window(saved + 64,  buf + 4 * 128 + 64, buf + 5 * 128, swindow, 0, 64);
        fmul_window(saved + 192, buf + 5 * 128 + 64, buf + 6 * 128, swindow, 0, 64);
        fmul_window(saved + 320, buf + 6 * 128 + 64, buf + 7 * 128, swindow, 0, 64);
        memcpy(saved + 448, buf + 7 * 128 + 64, 64 * sizeof(float));
    } else if (ws0 == HEAAC_LONG_START_SEQUENCE) {
        memcpy(saved,       buf + 512,          448 * sizeof(float));
        memcpy(saved + 448, buf + 7 * 128 + 64,  64 * sizeof(float));
    } else {
        memcpy(saved, buf + 512, 512 * sizeof(float));
    }
}

/* dsputil.c:3972-3981 float_to_int16_one */
int oracle_float_to_int16_one(float f)
{
    int32_t tmp;
    memcpy(&tmp, &f, 4);
    /* dsputil.c:3975-3980.  For a NEGATIVE float the reference's `0x43c0ffff - tmp` overflows int (undefined in C;
     * the two's-complement wrap is what a plain sub / sar pair computes): written out with unsigned arithmetic
     * here so that no optimiser can pick another answer.  Only reachable more than 385 full scales below zero. */
    if (tmp & 0xf0000)
        tmp = (int32_t)(0x43c0ffffu - (uint32_t)tmp) >> 31;
    return (int16_t)(tmp - 0x8000);
}

/* cvtps2dq (round to nearest even under the default MXCSR; NaN and |f| >= 2^31 give the "integer
 * indefinite" 0x80000000) followed by packssdw (signed saturation) */
int oracle_float_to_int16_sse2(float f)
{
    long v;
    if (!(fabsf(f) < 2147483648.0f))
        return -32768;
    v = lrintf(f);
    return v < -32768 ? -32768 : v > 32767 ? 32767 : (int)v;
}

/* dsputil.c:3989-4001 */
void oracle_float_to_int16_interleave(int16_t *dst, const float *const *src, long len, int channels, int sse2)
{
    long i, j;
    int c;
    if (channels == 2) {
        for (i = 0; i < len; i++) {
            dst[2 * i]     = (int16_t)(sse2 ? oracle_float_to_int16_sse2(src[0][i]) : oracle_float_to_int16_one(src[0][i]));
            dst[2 * i + 1] = (int16_t)(sse2 ? oracle_float_to_int16_sse2(src[1][i]) : oracle_float_to_int16_one(src[1][i]));
        }
    } else {
        for (c = 0; c < channels; c++)
            for (i = 0, j = c; i < len; i++, j += channels)
                dst[j] = (int16_t)(sse2 ? oracle_float_to_int16_sse2(src[c][i]) : oracle_float_to_int16_one(src[c][i]));
    }
}

static void store_pcm(void *pcm, int fmt, size_t frame, int nch, int len,
                      float *const *ch_ret)
{
    int c, i;
    if (fmt == HEAAC_PCM_F32_PLANAR) {
        float *p = (float *)pcm + frame * nch * len;
        for (c = 0; c < nch; c++)
            memcpy(p + (size_t)c * len, ch_ret[c], sizeof(float) * len);
    } else if (fmt == HEAAC_PCM_S16_INTERLEAVED_SSE2) {
        /* float_to_int16_interleave_sse2, x86/dsputil_mmx.c:2356-2372, 2405-2436 */
        int16_t *p = (int16_t *)pcm + frame * nch * len;
        for (i = 0; i < len; i++)
            for (c = 0; c < nch; c++)
                p[i * nch + c] = (int16_t)oracle_float_to_int16_sse2(ch_ret[c][i]);
    } else {
        /* dsputil.c:3989-4001 ff_float_to_int16_interleave_c */
        int16_t *p = (int16_t *)pcm + frame * nch * len;
        for (i = 0; i < len; i++)
            for (c = 0; c < nch; c++)
                p[i * nch + c] = (int16_t)oracle_float_to_int16_one(ch_ret[c][i]);
    }
}

/* exported for or_sbr.c */
void or_store_pcm(void *pcm, int fmt, size_t frame, int nch, int len, float *const *ch_ret)
{
    store_pcm(pcm, fmt, frame, nch, len, ch_ret);
}

/* spectral_to_sample for an AAC-LC SCE/CPE, aacdec.c:1903-1933 with sbr <= 0 */
int oracle_lc_decode_batch(int channels, const float *coeffs, const HeaacIcs *ics,
                           const float *state_in, float *state_out,
                           void *pcm, int pcm_format, size_t n)
{
    size_t f;
    int c;
    if (channels < 1 || channels > 2)
        return HEAAC_ERR_ARG;
    oracle_tables();
    for (f = 0; f < n; f++) {
        float ret[2][1024], saved[512];
        float *rp[2] = { ret[0], ret[1] };
        for (c = 0; c < channels; c++) {
            size_t u = f * channels + c;
            memcpy(saved, state_in + u * 512, sizeof(saved));
            /* add_bias of the C conversion or of the SIMD configuration (aacdec.c:573-581) */
            oracle_imdct_and_windowing(coeffs + u * 1024, &ics[u], saved, ret[c],
                                       pcm_format == HEAAC_PCM_S16_INTERLEAVED_SSE2 ? 0.0f : HEAAC_ADD_BIAS);
            memcpy(state_out + u * 512, saved, sizeof(saved));
        }
        store_pcm(pcm, pcm_format, f, channels, 1024, rp);
    }
    return 0;
}

/* apply_independent_coupling, aacdec.c:1849-1862 (len = 1024: no SBR), for the target channels whose
 * `on` flag is set, then (optionally) float_to_int16_interleave of the targets, dsputil.c:3989-4001 */
int oracle_couple_after_imdct_batch(int channels, float *pcm, const float *cce, const HeaacCoupling *cpl,
                                    int16_t *s16, size_t n)
{
    size_t f;
    int c, i;
    if (channels < 1 || channels > 2)
        return HEAAC_ERR_ARG;
    for (f = 0; f < n; f++) {
        const float *src = cce + f * 1024;
        for (c = 0; c < channels; c++) {
            float *dest = pcm + (f * channels + c) * 1024;
            const float gain = cpl[f].gain[c];
            const float bias = HEAAC_ADD_BIAS;
            if (cpl[f].on[c])
                for (i = 0; i < 1024; i++)
                    dest[i] += gain * (src[i] - bias);
            if (s16)
                for (i = 0; i < 1024; i++)
                    s16[(f * 1024 + i) * channels + c] = (int16_t)oracle_float_to_int16_one(dest[i]);
        }
    }
    return 0;
}
