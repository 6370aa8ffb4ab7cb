/* or_ps.c -- oracle: Parametric Stereo DSP (aacps.c:281-992).
 * TEST INFRASTRUCTURE (see oracle.h).
 */
#include <math.h>
#include <string.h>
#include "oracle.h"

#define PS_MAX_NUM_ENV 5
#define PS_MAX_NR_IIDICC 34
#define PS_MAX_DELAY 14
#define PS_AP_LINKS 3
#define numQMFSlots 32

/* aacpsdata.c:145-158, Tables 8.48 / 8.49 of ISO/IEC 14496-3 */
static const int8_t k_to_i_20[] = {
     1,  0,  0,  1,  2,  3,  4,  5,  6,  7,  8,  9, 10, 11, 12, 13, 14, 14, 15,
    15, 15, 16, 16, 16, 16, 17, 17, 17, 17, 17, 18, 18, 18, 18, 18, 18, 18, 18,
    18, 18, 18, 18, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19,
    19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19
};
static const int8_t k_to_i_34[] = {
     0,  1,  2,  3,  4,  5,  6,  6,  7,  2,  1,  0, 10, 10,  4,  5,  6,  7,  8,
     9, 10, 11, 12,  9, 14, 11, 12, 13, 14, 15, 16, 13, 16, 17, 18, 19, 20, 21,
    22, 22, 23, 23, 24, 24, 25, 25, 26, 26, 27, 27, 27, 28, 28, 28, 29, 29, 29,
    30, 30, 30, 31, 31, 31, 31, 32, 32, 32, 32, 33, 33, 33, 33, 33, 33, 33, 33,
    33, 33, 33, 33, 33, 33, 33, 33, 33, 33, 33, 33, 33, 33, 33
};
/* aacpsdata.c:160-163 */
static const float g1_Q2[] = {
    0.0f,  0.01899487526049f, 0.0f, -0.07293139167538f,
    0.0f,  0.30596630545168f, 0.5f
};

static const int NR_PAR_BANDS[]     = { 20, 34 };
static const int NR_BANDS[]         = { 71, 91 };
static const int DECAY_CUTOFF[]     = { 10, 32 };
static const int NR_ALLPASS_BANDS[] = { 30, 50 };
static const int SHORT_DELAY_BAND[] = { 42, 62 };
#define DECAY_SLOPE 0.05f

/* Working set shaped like PSContext (aacps.h:63-74). */
typedef struct {
    float in_buf[5][44][2];
    float delay[91][numQMFSlots + PS_MAX_DELAY][2];
    float ap_delay[50][PS_AP_LINKS][numQMFSlots + 5][2];
    float peak_decay_nrg[34], power_smooth[34], peak_decay_diff_smooth[34];
    float H11[2][PS_MAX_NUM_ENV + 1][34], H12[2][PS_MAX_NUM_ENV + 1][34];
    float H21[2][PS_MAX_NUM_ENV + 1][34], H22[2][PS_MAX_NUM_ENV + 1][34];
    int8_t opd_hist[34], ipd_hist[34];
} ps_ctx;

/* aacps.c:283-301 */
static void hybrid2_re(float (*in)[2], float (*out)[32][2], const float filter[7], int len, int reverse)
{
    int i, j;
    for (i = 0; i < len; i++, in++) {
        float re_in = filter[6] * in[6][0];
        float re_op = 0.0f;
        float im_in = filter[6] * in[6][1];
        float im_op = 0.0f;
        for (j = 0; j < 6; j += 2) {
            re_op += filter[j + 1] * (in[j + 1][0] + in[12 - j - 1][0]);
            im_op += filter[j + 1] * (in[j + 1][1] + in[12 - j - 1][1]);
        }
        out[ reverse][i][0] = re_in + re_op;
        out[ reverse][i][1] = im_in + im_op;
        out[!reverse][i][0] = re_in - re_op;
        out[!reverse][i][1] = im_in - im_op;
    }
}

/* common inner FIR of aacps.c:310-321 / :343-353 */
static void hybrid_fir(const float (*in)[2], const float (*filter)[2], float *o_re, float *o_im)
{
    int j;
    float sum_re = filter[6][0] * in[6][0], sum_im = filter[6][0] * in[6][1];
    for (j = 0; j < 6; j++) {
        float in0_re = in[j][0];
        float in0_im = in[j][1];
        float in1_re = in[12 - j][0];
        float in1_im = in[12 - j][1];
        sum_re += filter[j][0] * (in0_re + in1_re) - filter[j][1] * (in0_im - in1_im);
        sum_im += filter[j][0] * (in0_im + in1_im) + filter[j][1] * (in0_re - in1_re);
    }
    *o_re = sum_re;
    *o_im = sum_im;
}

/* aacps.c:303-336 */
static void hybrid6_cx(float (*in)[2], float (*out)[32][2], const float (*filter)[7][2], int len)
{
    int i, ssb;
    float temp[8][2];
    for (i = 0; i < len; i++, in++) {
        for (ssb = 0; ssb < 8; ssb++)
            hybrid_fir((const float (*)[2])in, filter[ssb], &temp[ssb][0], &temp[ssb][1]);
        out[0][i][0] = temp[6][0];
        out[0][i][1] = temp[6][1];
        out[1][i][0] = temp[7][0];
        out[1][i][1] = temp[7][1];
        out[2][i][0] = temp[0][0];
        out[2][i][1] = temp[0][1];
        out[3][i][0] = temp[1][0];
        out[3][i][1] = temp[1][1];
        out[4][i][0] = temp[2][0] + temp[5][0];
        out[4][i][1] = temp[2][1] + temp[5][1];
        out[5][i][0] = temp[3][0] + temp[4][0];
        out[5][i][1] = temp[3][1] + temp[4][1];
    }
}

/* aacps.c:338-357 */
static void hybrid4_8_12_cx(float (*in)[2], float (*out)[32][2], const float (*filter)[7][2], int N, int len)
{
    int i, ssb;
    for (i = 0; i < len; i++, in++)
        for (ssb = 0; ssb < N; ssb++)
            hybrid_fir((const float (*)[2])in, filter[ssb], &out[ssb][i][0], &out[ssb][i][1]);
}

/* aacps.c:359-395 */
static void hybrid_analysis(float out[91][32][2], float in[5][44][2], float L[2][38][64], int is34, int len)
{
    const or_tables *t = oracle_tables();
    int i, j;
    for (i = 0; i < 5; i++)
        for (j = 0; j < 38; j++) {
            in[i][j + 6][0] = L[0][j][i];
            in[i][j + 6][1] = L[1][j][i];
        }
    if (is34) {
        hybrid4_8_12_cx(in[0], out,      t->f34_0_12, 12, len);
        hybrid4_8_12_cx(in[1], out + 12, t->f34_1_8,   8, len);
        hybrid4_8_12_cx(in[2], out + 20, t->f34_2_4,   4, len);
        hybrid4_8_12_cx(in[3], out + 24, t->f34_2_4,   4, len);
        hybrid4_8_12_cx(in[4], out + 28, t->f34_2_4,   4, len);
        for (i = 0; i < 59; i++)
            for (j = 0; j < len; j++) {
                out[i + 32][j][0] = L[0][j][i + 5];
                out[i + 32][j][1] = L[1][j][i + 5];
            }
    } else {
        hybrid6_cx(in[0], out, t->f20_0_8, len);
        hybrid2_re(in[1], out + 6, g1_Q2, len, 1);
        hybrid2_re(in[2], out + 8, g1_Q2, len, 0);
        for (i = 0; i < 61; i++)
            for (j = 0; j < len; j++) {
                out[i + 10][j][0] = L[0][j][i + 3];
                out[i + 10][j][1] = L[1][j][i + 3];
            }
    }
    for (i = 0; i < 5; i++)
        memcpy(in[i], in[i] + 32, 6 * sizeof(in[i][0]));
}

/* aacps.c:397-445 */
static void hybrid_synthesis(float out[2][38][64], float in[91][32][2], int is34, int len)
{
    int i, n;
    if (is34) {
        for (n = 0; n < len; n++) {
            memset(out[0][n], 0, 5 * sizeof(out[0][n][0]));
            memset(out[1][n], 0, 5 * sizeof(out[1][n][0]));
            for (i = 0; i < 12; i++) {
                out[0][n][0] += in[i][n][0];
                out[1][n][0] += in[i][n][1];
            }
            for (i = 0; i < 8; i++) {
                out[0][n][1] += in[12 + i][n][0];
                out[1][n][1] += in[12 + i][n][1];
            }
            for (i = 0; i < 4; i++) {
                out[0][n][2] += in[20 + i][n][0];
                out[1][n][2] += in[20 + i][n][1];
                out[0][n][3] += in[24 + i][n][0];
                out[1][n][3] += in[24 + i][n][1];
                out[0][n][4] += in[28 + i][n][0];
                out[1][n][4] += in[28 + i][n][1];
            }
        }
        for (i = 0; i < 59; i++)
            for (n = 0; n < len; n++) {
                out[0][n][i + 5] = in[i + 32][n][0];
                out[1][n][i + 5] = in[i + 32][n][1];
            }
    } else {
        for (n = 0; n < len; n++) {
            out[0][n][0] = in[0][n][0] + in[1][n][0] + in[2][n][0] +
                           in[3][n][0] + in[4][n][0] + in[5][n][0];
            out[1][n][0] = in[0][n][1] + in[1][n][1] + in[2][n][1] +
                           in[3][n][1] + in[4][n][1] + in[5][n][1];
            out[0][n][1] = in[6][n][0] + in[7][n][0];
            out[1][n][1] = in[6][n][1] + in[7][n][1];
            out[0][n][2] = in[8][n][0] + in[9][n][0];
            out[1][n][2] = in[8][n][1] + in[9][n][1];
        }
        for (i = 0; i < 61; i++)
            for (n = 0; n < len; n++) {
                out[0][n][i + 3] = in[i + 10][n][0];
                out[1][n][i + 3] = in[i + 10][n][1];
            }
    }
}

/* ---- parameter index remapping, aacps.c:461-643, 756-792 ---- */
static void map_idx_10_to_20(int8_t *pm, const int8_t *par, int full)
{
    int b;
    if (full)
        b = 9;
    else {
        b = 4;
        pm[10] = 0;
    }
    for (; b >= 0; b--)
        pm[2 * b + 1] = pm[2 * b] = par[b];
}

static void map_idx_34_to_20(int8_t *pm, const int8_t *par, int full)
{
    pm[ 0] = (2 * par[ 0] +     par[ 1]) / 3;
    pm[ 1] = (    par[ 1] + 2 * par[ 2]) / 3;
    pm[ 2] = (2 * par[ 3] +     par[ 4]) / 3;
    pm[ 3] = (    par[ 4] + 2 * par[ 5]) / 3;
    pm[ 4] = (    par[ 6] +     par[ 7]) / 2;
    pm[ 5] = (    par[ 8] +     par[ 9]) / 2;
    pm[ 6] =      par[10];
    pm[ 7] =      par[11];
    pm[ 8] = (    par[12] +     par[13]) / 2;
    pm[ 9] = (    par[14] +     par[15]) / 2;
    pm[10] =      par[16];
    if (full) {
        pm[11] =  par[17];
        pm[12] =  par[18];
        pm[13] =  par[19];
        pm[14] = (par[20] + par[21]) / 2;
        pm[15] = (par[22] + par[23]) / 2;
        pm[16] = (par[24] + par[25]) / 2;
        pm[17] = (par[26] + par[27]) / 2;
        pm[18] = (par[28] + par[29] + par[30] + par[31]) / 4;
        pm[19] = (par[32] + par[33]) / 2;
    }
}

static void map_val_34_to_20(float par[34])
{
    par[ 0] = (2 * par[ 0] +     par[ 1]) * 0.33333333f;
    par[ 1] = (    par[ 1] + 2 * par[ 2]) * 0.33333333f;
    par[ 2] = (2 * par[ 3] +     par[ 4]) * 0.33333333f;
    par[ 3] = (    par[ 4] + 2 * par[ 5]) * 0.33333333f;
    par[ 4] = (    par[ 6] +     par[ 7]) * 0.5f;
    par[ 5] = (    par[ 8] +     par[ 9]) * 0.5f;
    par[ 6] =      par[10];
    par[ 7] =      par[11];
    par[ 8] = (    par[12] +     par[13]) * 0.5f;
    par[ 9] = (    par[14] +     par[15]) * 0.5f;
    par[10] =      par[16];
    par[11] =      par[17];
    par[12] =      par[18];
    par[13] =      par[19];
    par[14] = (    par[20] +     par[21]) * 0.5f;
    par[15] = (    par[22] +     par[23]) * 0.5f;
    par[16] = (    par[24] +     par[25]) * 0.5f;
    par[17] = (    par[26] +     par[27]) * 0.5f;
    par[18] = (    par[28] +     par[29] + par[30] + par[31]) * 0.25f;
    par[19] = (    par[32] +     par[33]) * 0.5f;
}

static const int8_t map_10_to_34[34] = {
    0, 0, 0, 1, 1, 1, 2, 2, 2, 2, 3, 3, 4, 4, 4, 4, 5, 5, 6, 6, 7, 7, 7, 7, 8, 8, 8, 8, 9, 9, 9, 9, 9, 9
};
static void map_idx_10_to_34(int8_t *pm, const int8_t *par, int full)
{
    int b;
    if (full) {
        for (b = 33; b >= 16; b--)
            pm[b] = par[map_10_to_34[b]];
    } else {
        pm[16] = 0;
    }
    for (b = 15; b >= 0; b--)
        pm[b] = par[map_10_to_34[b]];
}

static void map_idx_20_to_34(int8_t *pm, const int8_t *par, int full)
{
    if (full) {
        pm[33] = par[19]; pm[32] = par[19]; pm[31] = par[18]; pm[30] = par[18];
        pm[29] = par[18]; pm[28] = par[18]; pm[27] = par[17]; pm[26] = par[17];
        pm[25] = par[16]; pm[24] = par[16]; pm[23] = par[15]; pm[22] = par[15];
        pm[21] = par[14]; pm[20] = par[14]; pm[19] = par[13]; pm[18] = par[12];
        pm[17] = par[11];
    }
    pm[16] = par[10]; pm[15] = par[ 9]; pm[14] = par[ 9]; pm[13] = par[ 8];
    pm[12] = par[ 8]; pm[11] = par[ 7]; pm[10] = par[ 6]; pm[ 9] = par[ 5];
    pm[ 8] = par[ 5]; pm[ 7] = par[ 4]; pm[ 6] = par[ 4]; pm[ 5] = par[ 3];
    pm[ 4] = (par[2] + par[3]) / 2;
    pm[ 3] = par[ 2]; pm[ 2] = par[ 1];
    pm[ 1] = (par[0] + par[1]) / 2;
    pm[ 0] = par[ 0];
}

static void map_val_20_to_34(float par[34])
{
    par[33] = par[19]; par[32] = par[19]; par[31] = par[18]; par[30] = par[18];
    par[29] = par[18]; par[28] = par[18]; par[27] = par[17]; par[26] = par[17];
    par[25] = par[16]; par[24] = par[16]; par[23] = par[15]; par[22] = par[15];
    par[21] = par[14]; par[20] = par[14]; par[19] = par[13]; par[18] = par[12];
    par[17] = par[11]; par[16] = par[10]; par[15] = par[ 9]; par[14] = par[ 9];
    par[13] = par[ 8]; par[12] = par[ 8]; par[11] = par[ 7]; par[10] = par[ 6];
    par[ 9] = par[ 5]; par[ 8] = par[ 5]; par[ 7] = par[ 4]; par[ 6] = par[ 4];
    par[ 5] = par[ 3];
    par[ 4] = (par[2] + par[3]) * 0.5f;
    par[ 3] = par[ 2];
    par[ 2] = par[ 1];
    par[ 1] = (par[0] + par[1]) * 0.5f;
    par[ 0] = par[ 0];
}

/* remap34 / remap20, aacps.c:756-792.  `src` rows have `stride` entries. */
static void remap(int to34, int8_t (*dst)[34], const int8_t *src, int stride,
                  int num_par, int num_env, int full)
{
    int e, b;
    for (e = 0; e < num_env; e++) {
        const int8_t *par = src + e * stride;
        if (to34) {
            if (num_par == 20 || num_par == 11)
                map_idx_20_to_34(dst[e], par, full);
            else if (num_par == 10 || num_par == 5)
                map_idx_10_to_34(dst[e], par, full);
            else
                for (b = 0; b < stride; b++) dst[e][b] = par[b];
        } else {
            if (num_par == 34 || num_par == 17)
                map_idx_34_to_20(dst[e], par, full);
            else if (num_par == 10 || num_par == 5)
                map_idx_10_to_20(dst[e], par, full);
            else
                for (b = 0; b < stride; b++) dst[e][b] = par[b];
        }
    }
}

/* aacps.c:645-754 */
static void decorrelation(ps_ctx *ps, float (*out)[32][2], const float (*s)[32][2], int is34, int is34_old)
{
    const or_tables *t = oracle_tables();
    static __thread float power[34][32];          /* (thread-local: the CPU baseline runs the oracle on all cores) */
    static __thread float transient_gain[34][32];
    float *peak_decay_nrg = ps->peak_decay_nrg;
    float *power_smooth = ps->power_smooth;
    float *peak_decay_diff_smooth = ps->peak_decay_diff_smooth;
    float (*delay)[numQMFSlots + PS_MAX_DELAY][2] = ps->delay;
    float (*ap_delay)[PS_AP_LINKS][numQMFSlots + 5][2] = ps->ap_delay;
    const int8_t *k_to_i = is34 ? k_to_i_34 : k_to_i_20;
    const float peak_decay_factor = 0.76592833836465f;
    const float transient_impact  = 1.5f;
    const float a_smooth          = 0.25f;
    int i, k, m, n;
    const int n0 = 0, nL = 32;
    static const int link_delay[] = { 3, 4, 5 };
    static const float a[] = { 0.65143905753106f, 0.56471812200776f, 0.48954165955695f };

    memset(power, 0, sizeof(power));
    if (is34 != is34_old) {
        memset(ps->peak_decay_nrg,         0, sizeof(ps->peak_decay_nrg));
        memset(ps->power_smooth,           0, sizeof(ps->power_smooth));
        memset(ps->peak_decay_diff_smooth, 0, sizeof(ps->peak_decay_diff_smooth));
        memset(ps->delay,                  0, sizeof(ps->delay));
        memset(ps->ap_delay,               0, sizeof(ps->ap_delay));
    }

    for (n = n0; n < nL; n++)
        for (k = 0; k < NR_BANDS[is34]; k++) {
            int i = k_to_i[k];
            power[i][n] += s[k][n][0] * s[k][n][0] + s[k][n][1] * s[k][n][1];
        }

    for (i = 0; i < NR_PAR_BANDS[is34]; i++)
        for (n = n0; n < nL; n++) {
            float decayed_peak = peak_decay_factor * peak_decay_nrg[i];
            float denom;
            peak_decay_nrg[i] = decayed_peak > power[i][n] ? decayed_peak : power[i][n];
            power_smooth[i] += a_smooth * (power[i][n] - power_smooth[i]);
            peak_decay_diff_smooth[i] += a_smooth * (peak_decay_nrg[i] - power[i][n] - peak_decay_diff_smooth[i]);
            denom = transient_impact * peak_decay_diff_smooth[i];
            transient_gain[i][n] = (denom > power_smooth[i]) ? power_smooth[i] / denom : 1.0f;
        }

    for (k = 0; k < NR_ALLPASS_BANDS[is34]; k++) {
        int b = k_to_i[k];
        float g_decay_slope = 1.f - DECAY_SLOPE * (k - DECAY_CUTOFF[is34]);
        float ag[PS_AP_LINKS];
        /* av_clipf(x, 0, 1) */
        if (g_decay_slope < 0.f) g_decay_slope = 0.f;
        else if (g_decay_slope > 1.f) g_decay_slope = 1.f;
        memcpy(delay[k], delay[k] + nL, PS_MAX_DELAY * sizeof(delay[k][0]));
        memcpy(delay[k] + PS_MAX_DELAY, s[k], numQMFSlots * sizeof(delay[k][0]));
        for (m = 0; m < PS_AP_LINKS; m++) {
            memcpy(ap_delay[k][m], ap_delay[k][m] + numQMFSlots, 5 * sizeof(ap_delay[k][m][0]));
            ag[m] = a[m] * g_decay_slope;
        }
        for (n = n0; n < nL; n++) {
            float in_re = delay[k][n + PS_MAX_DELAY - 2][0] * t->phi_fract[is34][k][0] -
                          delay[k][n + PS_MAX_DELAY - 2][1] * t->phi_fract[is34][k][1];
            float in_im = delay[k][n + PS_MAX_DELAY - 2][0] * t->phi_fract[is34][k][1] +
                          delay[k][n + PS_MAX_DELAY - 2][1] * t->phi_fract[is34][k][0];
            for (m = 0; m < PS_AP_LINKS; m++) {
                float a_re                = ag[m] * in_re;
                float a_im                = ag[m] * in_im;
                float link_delay_re       = ap_delay[k][m][n + 5 - link_delay[m]][0];
                float link_delay_im       = ap_delay[k][m][n + 5 - link_delay[m]][1];
                float fractional_delay_re = t->Q_fract_allpass[is34][k][m][0];
                float fractional_delay_im = t->Q_fract_allpass[is34][k][m][1];
                ap_delay[k][m][n + 5][0] = in_re;
                ap_delay[k][m][n + 5][1] = in_im;
                in_re = link_delay_re * fractional_delay_re - link_delay_im * fractional_delay_im - a_re;
                in_im = link_delay_re * fractional_delay_im + link_delay_im * fractional_delay_re - a_im;
                ap_delay[k][m][n + 5][0] += ag[m] * in_re;
                ap_delay[k][m][n + 5][1] += ag[m] * in_im;
            }
            out[k][n][0] = transient_gain[b][n] * in_re;
            out[k][n][1] = transient_gain[b][n] * in_im;
        }
    }
    for (; k < SHORT_DELAY_BAND[is34]; k++) {
        memcpy(delay[k], delay[k] + nL, PS_MAX_DELAY * sizeof(delay[k][0]));
        memcpy(delay[k] + PS_MAX_DELAY, s[k], numQMFSlots * sizeof(delay[k][0]));
        for (n = n0; n < nL; n++) {
            out[k][n][0] = transient_gain[k_to_i[k]][n] * delay[k][n + PS_MAX_DELAY - 14][0];
            out[k][n][1] = transient_gain[k_to_i[k]][n] * delay[k][n + PS_MAX_DELAY - 14][1];
        }
    }
    for (; k < NR_BANDS[is34]; k++) {
        memcpy(delay[k], delay[k] + nL, PS_MAX_DELAY * sizeof(delay[k][0]));
        memcpy(delay[k] + PS_MAX_DELAY, s[k], numQMFSlots * sizeof(delay[k][0]));
        for (n = n0; n < nL; n++) {
            out[k][n][0] = transient_gain[k_to_i[k]][n] * delay[k][n + PS_MAX_DELAY - 1][0];
            out[k][n][1] = transient_gain[k_to_i[k]][n] * delay[k][n + PS_MAX_DELAY - 1][1];
        }
    }
}

/* aacps.c:794-971 (PS_BASELINE == 0) */
static void stereo_processing(const HeaacPsFrame *p, ps_ctx *ps, float (*l)[32][2], float (*r)[32][2], int is34)
{
    const or_tables *t = oracle_tables();
    int e, b, k, n;
    float (*H11)[PS_MAX_NUM_ENV + 1][34] = ps->H11;
    float (*H12)[PS_MAX_NUM_ENV + 1][34] = ps->H12;
    float (*H21)[PS_MAX_NUM_ENV + 1][34] = ps->H21;
    float (*H22)[PS_MAX_NUM_ENV + 1][34] = ps->H22;
    int8_t *opd_hist = ps->opd_hist;
    int8_t *ipd_hist = ps->ipd_hist;
    int8_t iid_mapped[PS_MAX_NUM_ENV][34];
    int8_t icc_mapped[PS_MAX_NUM_ENV][34];
    int8_t ipd_mapped[PS_MAX_NUM_ENV][34];
    int8_t opd_mapped[PS_MAX_NUM_ENV][34];
    const int8_t *k_to_i = is34 ? k_to_i_34 : k_to_i_20;
    const float (*H_LUT)[8][4] = (p->icc_mode < 3) ? t->HA : t->HB;

    /* H[.][0] <- H[.][num_env_old]: the state record already holds that row in
     * row 0 (see or_ps_apply), so the copy at aacps.c:818-825 is a no-op here. */
    /* The reference leaves these four arrays uninitialised (aacps.c:804-807).  One record reads entries it
     * never wrote: 17 phase parameters on the 20-band grid (enable_iid off, nr_ipdopd_par left at 17 by an
     * earlier header): remap20 with full == 0 writes entries 0..10, the mixing loop reads up to 16 (:863).
     * Defined here -- and in the kernel -- as zero. */
    memset(ipd_mapped, 0, sizeof(ipd_mapped));
    memset(opd_mapped, 0, sizeof(opd_mapped));
    memset(iid_mapped, 0, sizeof(iid_mapped));
    memset(icc_mapped, 0, sizeof(icc_mapped));
    remap(is34, iid_mapped, &p->iid_par[0][0], 34, p->nr_iid_par, p->num_env, 1);
    remap(is34, icc_mapped, &p->icc_par[0][0], 34, p->nr_icc_par, p->num_env, 1);
    if (p->enable_ipdopd) {
        remap(is34, ipd_mapped, &p->ipd_par[0][0], 17, p->nr_ipdopd_par, p->num_env, 0);
        remap(is34, opd_mapped, &p->opd_par[0][0], 17, p->nr_ipdopd_par, p->num_env, 0);
    }
    if (is34 && !p->is34bands_old) {
        map_val_20_to_34(H11[0][0]); map_val_20_to_34(H11[1][0]);
        map_val_20_to_34(H12[0][0]); map_val_20_to_34(H12[1][0]);
        map_val_20_to_34(H21[0][0]); map_val_20_to_34(H21[1][0]);
        map_val_20_to_34(H22[0][0]); map_val_20_to_34(H22[1][0]);
        memset(ipd_hist, 0, 17); memset(opd_hist, 0, 17);   /* ipdopd_reset */
    } else if (!is34 && p->is34bands_old) {
        map_val_34_to_20(H11[0][0]); map_val_34_to_20(H11[1][0]);
        map_val_34_to_20(H12[0][0]); map_val_34_to_20(H12[1][0]);
        map_val_34_to_20(H21[0][0]); map_val_34_to_20(H21[1][0]);
        map_val_34_to_20(H22[0][0]); map_val_34_to_20(H22[1][0]);
        memset(ipd_hist, 0, 17); memset(opd_hist, 0, 17);
    }

    for (e = 0; e < p->num_env; e++) {
        for (b = 0; b < NR_PAR_BANDS[is34]; b++) {
            float h11, h12, h21, h22;
            h11 = H_LUT[iid_mapped[e][b] + 7 + 23 * p->iid_quant][icc_mapped[e][b]][0];
            h12 = H_LUT[iid_mapped[e][b] + 7 + 23 * p->iid_quant][icc_mapped[e][b]][1];
            h21 = H_LUT[iid_mapped[e][b] + 7 + 23 * p->iid_quant][icc_mapped[e][b]][2];
            h22 = H_LUT[iid_mapped[e][b] + 7 + 23 * p->iid_quant][icc_mapped[e][b]][3];
            if (p->enable_ipdopd && b < p->nr_ipdopd_par) {
                float h11i, h12i, h21i, h22i;
                float ipd_adj_re, ipd_adj_im;
                int opd_idx = opd_hist[b] * 8 + opd_mapped[e][b];
                int ipd_idx = ipd_hist[b] * 8 + ipd_mapped[e][b];
                float opd_re = t->pd_re_smooth[opd_idx];
                float opd_im = t->pd_im_smooth[opd_idx];
                float ipd_re = t->pd_re_smooth[ipd_idx];
                float ipd_im = t->pd_im_smooth[ipd_idx];
                opd_hist[b] = opd_idx & 0x3F;
                ipd_hist[b] = ipd_idx & 0x3F;

                ipd_adj_re = opd_re * ipd_re + opd_im * ipd_im;
                ipd_adj_im = opd_im * ipd_re - opd_re * ipd_im;
                h11i = h11 * opd_im;
                h11  = h11 * opd_re;
                h12i = h12 * ipd_adj_im;
                h12  = h12 * ipd_adj_re;
                h21i = h21 * opd_im;
                h21  = h21 * opd_re;
                h22i = h22 * ipd_adj_im;
                h22  = h22 * ipd_adj_re;
                H11[1][e + 1][b] = h11i;
                H12[1][e + 1][b] = h12i;
                H21[1][e + 1][b] = h21i;
                H22[1][e + 1][b] = h22i;
            }
            H11[0][e + 1][b] = h11;
            H12[0][e + 1][b] = h12;
            H21[0][e + 1][b] = h21;
            H22[0][e + 1][b] = h22;
        }
        for (k = 0; k < NR_BANDS[is34]; k++) {
            float h11r, h12r, h21r, h22r;
            float h11i = 0, h12i = 0, h21i = 0, h22i = 0;
            float h11r_step, h12r_step, h21r_step, h22r_step;
            float h11i_step = 0, h12i_step = 0, h21i_step = 0, h22i_step = 0;
            int start = p->border_position[e];
            int stop  = p->border_position[e + 1];
            float width = 1.f / (stop - start);
            b = k_to_i[k];
            h11r = H11[0][e][b];
            h12r = H12[0][e][b];
            h21r = H21[0][e][b];
            h22r = H22[0][e][b];
            if (p->enable_ipdopd) {
                if ((is34 && k <= 13 && k >= 9) || (!is34 && k <= 1)) {
                    h11i = -H11[1][e][b];
                    h12i = -H12[1][e][b];
                    h21i = -H21[1][e][b];
                    h22i = -H22[1][e][b];
                } else {
                    h11i = H11[1][e][b];
                    h12i = H12[1][e][b];
                    h21i = H21[1][e][b];
                    h22i = H22[1][e][b];
                }
            }
            h11r_step = (H11[0][e + 1][b] - h11r) * width;
            h12r_step = (H12[0][e + 1][b] - h12r) * width;
            h21r_step = (H21[0][e + 1][b] - h21r) * width;
            h22r_step = (H22[0][e + 1][b] - h22r) * width;
            if (p->enable_ipdopd) {
                h11i_step = (H11[1][e + 1][b] - h11i) * width;
                h12i_step = (H12[1][e + 1][b] - h12i) * width;
                h21i_step = (H21[1][e + 1][b] - h21i) * width;
                h22i_step = (H22[1][e + 1][b] - h22i) * width;
            }
            for (n = start + 1; n <= stop; n++) {
                float l_re = l[k][n][0];
                float l_im = l[k][n][1];
                float r_re = r[k][n][0];
                float r_im = r[k][n][1];
                h11r += h11r_step;
                h12r += h12r_step;
                h21r += h21r_step;
                h22r += h22r_step;
                if (p->enable_ipdopd) {
                    h11i += h11i_step;
                    h12i += h12i_step;
                    h21i += h21i_step;
                    h22i += h22i_step;
                    l[k][n][0] = h11r * l_re + h21r * r_re - h11i * l_im - h21i * r_im;
                    l[k][n][1] = h11r * l_im + h21r * r_im + h11i * l_re + h21i * r_re;
                    r[k][n][0] = h12r * l_re + h22r * r_re - h12i * l_im - h22i * r_im;
                    r[k][n][1] = h12r * l_im + h22r * r_im + h12i * l_re + h22i * r_re;
                } else {
                    l[k][n][0] = h11r * l_re + h21r * r_re;
                    l[k][n][1] = h11r * l_im + h21r * r_im;
                    r[k][n][0] = h12r * l_re + h22r * r_re;
                    r[k][n][1] = h12r * l_im + h22r * r_im;
                }
            }
        }
    }
}

/* ff_ps_apply, aacps.c:973-992, on the packed state record (in place). */
void or_ps_apply(const HeaacPsFrame *p, float *st, float L[2][38][64], float R[2][38][64], int top)
{
    static __thread ps_ctx ps;
    static __thread float Lbuf[91][32][2], Rbuf[91][32][2];
    const int len = 32;
    const int is34 = p->is34bands;
    int i, k, m, j, c;
    float *H[4][2];

    memset(&ps, 0, sizeof(ps));
    /* unpack */
    for (i = 0; i < 5; i++)
        memcpy(ps.in_buf[i], st + HEAAC_PS_INBUF + i * 12, 12 * sizeof(float));
    /* state layout is band-fastest: delay[j][k][re,im], ap_delay[m][j][k][re,im] */
    for (k = 0; k < 91; k++)
        for (j = 0; j < 14; j++) {
            ps.delay[k][32 + j][0] = st[HEAAC_PS_DELAY + (j * 91 + k) * 2];
            ps.delay[k][32 + j][1] = st[HEAAC_PS_DELAY + (j * 91 + k) * 2 + 1];
        }
    for (k = 0; k < 50; k++)
        for (m = 0; m < 3; m++)
            for (j = 0; j < 5; j++) {
                ps.ap_delay[k][m][32 + j][0] = st[HEAAC_PS_APDELAY + ((m * 5 + j) * 50 + k) * 2];
                ps.ap_delay[k][m][32 + j][1] = st[HEAAC_PS_APDELAY + ((m * 5 + j) * 50 + k) * 2 + 1];
            }
    memcpy(ps.peak_decay_nrg, st + HEAAC_PS_PEAK, 34 * sizeof(float));
    memcpy(ps.power_smooth, st + HEAAC_PS_PSMOOTH, 34 * sizeof(float));
    memcpy(ps.peak_decay_diff_smooth, st + HEAAC_PS_PDIFF, 34 * sizeof(float));
    H[0][0] = ps.H11[0][0]; H[0][1] = ps.H11[1][0];
    H[1][0] = ps.H12[0][0]; H[1][1] = ps.H12[1][0];
    H[2][0] = ps.H21[0][0]; H[2][1] = ps.H21[1][0];
    H[3][0] = ps.H22[0][0]; H[3][1] = ps.H22[1][0];
    for (j = 0; j < 4; j++)
        for (c = 0; c < 2; c++)
            memcpy(H[j][c], st + HEAAC_PS_H + (j * 2 + c) * 34, 34 * sizeof(float));
    memcpy(ps.opd_hist, (const char *)(st + HEAAC_PS_HIST), 34);
    memcpy(ps.ipd_hist, (const char *)(st + HEAAC_PS_HIST) + 34, 34);

    /* aacps.c:980-983 */
    top += NR_BANDS[is34] - 64;
    memset(ps.delay + top, 0, (NR_BANDS[is34] - top) * sizeof(ps.delay[0]));
    if (top < NR_ALLPASS_BANDS[is34])
        memset(ps.ap_delay + top, 0, (NR_ALLPASS_BANDS[is34] - top) * sizeof(ps.ap_delay[0]));

    hybrid_analysis(Lbuf, ps.in_buf, L, is34, len);
    decorrelation(&ps, Rbuf, (const float (*)[32][2])Lbuf, is34, p->is34bands_old);
    stereo_processing(p, &ps, Lbuf, Rbuf, is34);
    hybrid_synthesis(L, Lbuf, is34, len);
    hybrid_synthesis(R, Rbuf, is34, len);

    /* pack: the rows the next frame will read */
    for (i = 0; i < 5; i++)
        memcpy(st + HEAAC_PS_INBUF + i * 12, ps.in_buf[i], 12 * sizeof(float));
    for (k = 0; k < 91; k++)
        for (j = 0; j < 14; j++) {
            st[HEAAC_PS_DELAY + (j * 91 + k) * 2]     = ps.delay[k][32 + j][0];
            st[HEAAC_PS_DELAY + (j * 91 + k) * 2 + 1] = ps.delay[k][32 + j][1];
        }
    for (k = 0; k < 50; k++)
        for (m = 0; m < 3; m++)
            for (j = 0; j < 5; j++) {
                st[HEAAC_PS_APDELAY + ((m * 5 + j) * 50 + k) * 2]     = ps.ap_delay[k][m][32 + j][0];
                st[HEAAC_PS_APDELAY + ((m * 5 + j) * 50 + k) * 2 + 1] = ps.ap_delay[k][m][32 + j][1];
            }
    memcpy(st + HEAAC_PS_PEAK, ps.peak_decay_nrg, 34 * sizeof(float));
    memcpy(st + HEAAC_PS_PSMOOTH, ps.power_smooth, 34 * sizeof(float));
    memcpy(st + HEAAC_PS_PDIFF, ps.peak_decay_diff_smooth, 34 * sizeof(float));
    H[0][0] = ps.H11[0][p->num_env]; H[0][1] = ps.H11[1][p->num_env];
    H[1][0] = ps.H12[0][p->num_env]; H[1][1] = ps.H12[1][p->num_env];
    H[2][0] = ps.H21[0][p->num_env]; H[2][1] = ps.H21[1][p->num_env];
    H[3][0] = ps.H22[0][p->num_env]; H[3][1] = ps.H22[1][p->num_env];
    /* Imaginary rows exist only while IPD/OPD is enabled; the reference keeps
     * whatever an older frame left in them, which a minimal state record
     * cannot reproduce.  Defined here (and in the HIP path) as: pass through
     * unchanged while enable_ipdopd == 0.  See DESIGN.md "state semantics". */
    for (j = 0; j < 4; j++)
        for (c = 0; c < (p->enable_ipdopd ? 2 : 1); c++)
            memcpy(st + HEAAC_PS_H + (j * 2 + c) * 34, H[j][c], 34 * sizeof(float));
    memcpy((char *)(st + HEAAC_PS_HIST), ps.opd_hist, 34);
    memcpy((char *)(st + HEAAC_PS_HIST) + 34, ps.ipd_hist, 34);
}
