/* or_sbr.c -- oracle: SBR DSP stages (aacsbr.c:1088-1771) and the whole-frame
 * HE-AAC drivers.  TEST INFRASTRUCTURE (see oracle.h).
 */
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define FFMIN(a,b) ((a) > (b) ? (b) : (a))
#define FFMAX(a,b) ((a) > (b) ? (a) : (b))

#define ENV_ADJ_OFFSET 2          /* aacsbr.c:39 */
#define NOISE_FLOOR_OFFSET 6.0f   /* aacsbr.c:40 */

void or_store_pcm(void *pcm, int fmt, size_t frame, int nch, int len, float *const *ch_ret);
void or_ps_apply(const HeaacPsFrame *p, float *ps_state, float L[2][38][64], float R[2][38][64], int top);

/* Per-channel working set, shaped like SBRData (sbr.h:60-106). */
typedef struct {
    float xbuf[1312];
    float W[2][32][32][2];
    float Y[2][38][64][2];
    float g_temp[42][48], q_temp[42][48];
    uint8_t s_indexmapped[8][48];
    float env_facs[6][48];
    float noise_facs[3][5];
    float bw_array[5];
    unsigned f_indexnoise, f_indexsine;
} sbr_ch;

/* Shared scratch, shaped like SpectralBandReplication (sbr.h:140-162). */
typedef struct {
    float X_low[32][40][2];
    float X_high[64][40][2];
    float X[2][2][38][64];
    float alpha0[64][2], alpha1[64][2];
    float e_origmapped[7][48], q_mapped[7][48];
    uint8_t s_mapped[7][48];
    float e_curr[7][48], q_m[7][48], s_m[7][48], gain[7][48];
} sbr_scratch;

/* ------------------------------------------------------------------ */
/* a10 sbr_dequant, aacsbr.c:1089-1128                                  */
/* ------------------------------------------------------------------ */
static void dequant(const HeaacSbrFrame *fr, const HeaacSbrHeader *h, int cpe, sbr_ch *d0, sbr_ch *d1)
{
    int k, e, ch;
    const HeaacSbrChannel *c0 = &fr->ch[0];
    /* the parser leaves the integers in the float arrays */
    for (ch = 0; ch < (cpe ? 2 : 1); ch++) {
        sbr_ch *d = ch ? d1 : d0;
        const HeaacSbrChannel *c = &fr->ch[ch];
        for (e = 1; e <= 5; e++)
            for (k = 0; k < 48; k++)
                d->env_facs[e][k] = c->env_facs_q[e - 1][k];
        for (e = 1; e <= 2; e++)
            for (k = 0; k < 5; k++)
                d->noise_facs[e][k] = c->noise_facs_q[e - 1][k];
    }
    if (cpe && fr->bs_coupling) {
        float alpha      = c0->bs_amp_res ?  1.0f :  0.5f;
        float pan_offset = c0->bs_amp_res ? 12.0f : 24.0f;
        for (e = 1; e <= c0->bs_num_env; e++) {
            for (k = 0; k < h->n[c0->bs_freq_res[e]]; k++) {
                float temp1 = exp2f(d0->env_facs[e][k] * alpha + 7.0f);
                float temp2 = exp2f((pan_offset - d1->env_facs[e][k]) * alpha);
                float fac   = temp1 / (1.0f + temp2);
                d0->env_facs[e][k] = fac;
                d1->env_facs[e][k] = fac * temp2;
            }
        }
        for (e = 1; e <= c0->bs_num_noise; e++) {
            for (k = 0; k < h->n_q; k++) {
                float temp1 = exp2f(NOISE_FLOOR_OFFSET - d0->noise_facs[e][k] + 1);
                float temp2 = exp2f(12 - d1->noise_facs[e][k]);
                float fac   = temp1 / (1.0f + temp2);
                d0->noise_facs[e][k] = fac;
                d1->noise_facs[e][k] = fac * temp2;
            }
        }
    } else {
        for (ch = 0; ch < (cpe ? 2 : 1); ch++) {
            sbr_ch *d = ch ? d1 : d0;
            const HeaacSbrChannel *c = &fr->ch[ch];
            float alpha = c->bs_amp_res ? 1.0f : 0.5f;
            for (e = 1; e <= c->bs_num_env; e++)
                for (k = 0; k < h->n[c->bs_freq_res[e]]; k++)
                    d->env_facs[e][k] = exp2f(alpha * d->env_facs[e][k] + 6.0f);
            for (e = 1; e <= c->bs_num_noise; e++)
                for (k = 0; k < h->n_q; k++)
                    d->noise_facs[e][k] = exp2f(NOISE_FLOOR_OFFSET - d->noise_facs[e][k]);
        }
    }
}

/* ------------------------------------------------------------------ */
/* a11 sbr_qmf_analysis, aacsbr.c:1136-1169                             */
/* ------------------------------------------------------------------ */
static void qmf_analysis(const float *in, float *x, float W[2][32][32][2], float scale)
{
    const or_tables *t = oracle_tables();
    float z[320];
    int i, k;
    memcpy(W[0], W[1], sizeof(W[0]));
    memcpy(x, x + 1024, (320 - 32) * sizeof(x[0]));
    if (scale != 1.0f) {
        for (i = 0; i < 1024; i++)          /* vector_fmul_scalar_c */
            x[288 + i] = in[i] * scale;
    } else
        memcpy(x + 288, in, 1024 * sizeof(*x));
    for (i = 0; i < 32; i++) {
        for (k = 0; k < 320; k++)           /* vector_fmul_reverse_c */
            z[k] = t->qmf_ds[k] * x[319 - k];
        for (k = 0; k < 64; k++) {
            float f = z[k] + z[k + 64] + z[k + 128] + z[k + 192] + z[k + 256];
            z[k] = f;
        }
        z[64] = z[0];
        for (k = 1; k < 32; k++) {
            z[64 + 2 * k - 1] =  z[k];
            z[64 + 2 * k    ] = -z[64 - k];
        }
        z[64 + 63] = z[32];
        oracle_imdct_half(3, z, z + 64);
        for (k = 0; k < 32; k++) {
            W[1][i][k][0] = -z[63 - k];
            W[1][i][k][1] = z[k];
        }
        x += 32;
    }
}

void oracle_qmf_analysis(const float *in, float *xhist, float *Wout, float scale)
{
    static __thread float x[1312];
    static __thread float W[2][32][32][2];
    memset(W, 0, sizeof(W));
    memcpy(x + 1024, xhist, 288 * sizeof(float));
    qmf_analysis(in, x, W, scale);
    memcpy(xhist, x + 1024, 288 * sizeof(float));
    memcpy(Wout, W[1], sizeof(W[1]));
}

/* ------------------------------------------------------------------ */
/* a20 sbr_qmf_synthesis (div = 0), aacsbr.c:1175-1230                  */
/* The reference's 2304-float ring (v_off walking down by 128 per slot,  */
/* wrap-copy at 0) is restated on a linear buffer: slot i's 128 new      */
/* values sit at vb + (31-i)*128, the 1152 history values (newest first) */
/* follow at vb + 32*128.  v + off in the reference == vs + off here.    */
/* ------------------------------------------------------------------ */
static void qmf_synthesis(float *out, float X[2][38][64], float *vstate, float bias, float scale)
{
    const or_tables *t = oracle_tables();
    static const int voff[10] = { 0, 192, 256, 448, 512, 704, 768, 960, 1024, 1216 };
    float vb[41 * 128];
    float mdct_buf[2][64];
    int i, n, j;
    int scale_and_bias = scale != 1.0f || bias != 0.0f;
    memcpy(vb + 32 * 128, vstate, 1152 * sizeof(float));
    for (i = 0; i < 32; i++) {
        float *v = vb + (31 - i) * 128;
        for (n = 1; n < 64; n += 2)
            X[1][i][n] = -X[1][i][n];
        oracle_imdct_half(2, mdct_buf[0], X[0][i]);
        oracle_imdct_half(2, mdct_buf[1], X[1][i]);
        for (n = 0; n < 64; n++) {
            v[      n] = -mdct_buf[0][63 - n] + mdct_buf[1][n];
            v[127 - n] =  mdct_buf[0][63 - n] + mdct_buf[1][n];
        }
        for (n = 0; n < 64; n++)            /* vector_fmul_add(out, v, w, zero64) */
            out[n] = v[n] * t->qmf_us[n] + 0.0f;
        for (j = 1; j < 10; j++)
            for (n = 0; n < 64; n++)
                out[n] = v[voff[j] + n] * t->qmf_us[64 * j + n] + out[n];
        if (scale_and_bias)
            for (n = 0; n < 64; n++)
                out[n] = out[n] * scale + bias;
        out += 64;
    }
    memcpy(vstate, vb, 1152 * sizeof(float));
}

void oracle_qmf_synthesis(const float *Xin, float *v, float *out, float scale, float bias)
{
    static __thread float X[2][38][64];
    int p, i;
    memset(X, 0, sizeof(X));
    for (p = 0; p < 2; p++)
        for (i = 0; i < 32; i++)
            memcpy(X[p][i], Xin + (p * 32 + i) * 64, 64 * sizeof(float));
    qmf_synthesis(out, X, v, bias, scale);
}

/* Downsampled synthesis QMF bank: sbr_qmf_synthesis with div = 1 (aacsbr.c:1175-1230): 32 output
 * samples per slot from one 128-point IMDCT, window sbr_qmf_window_ds, ring slots of 64 values.
 * Same linear restatement as above: slot i at vb + (31-i)*64, 576 history values behind. */
void oracle_qmf_synthesis_ds(const float *Xin, float *vstate, float *out, float scale, float bias)
{
    const or_tables *t = oracle_tables();
    static const int voff[10] = { 0, 96, 128, 224, 256, 352, 384, 480, 512, 608 };
    float vb[41 * 64];
    float mdct_buf[64], in[64];
    int i, n, j;
    int scale_and_bias = scale != 1.0f || bias != 0.0f;
    memcpy(vb + 32 * 64, vstate, 576 * sizeof(float));
    for (i = 0; i < 32; i++) {
        const float *X0 = Xin + i * 64, *X1 = Xin + (32 + i) * 64;
        float *v = vb + (31 - i) * 64;
        for (n = 0; n < 32; n++) {
            in[n]      = -X0[n];
            in[32 + n] =  X1[31 - n];
        }
        oracle_imdct_half(2, mdct_buf, in);
        for (n = 0; n < 32; n++) {
            v[     n] =  mdct_buf[63 - 2 * n];
            v[63 - n] = -mdct_buf[62 - 2 * n];
        }
        for (n = 0; n < 32; n++)
            out[n] = v[n] * t->qmf_ds[n] + 0.0f;
        for (j = 1; j < 10; j++)
            for (n = 0; n < 32; n++)
                out[n] = v[voff[j] + n] * t->qmf_ds[32 * j + n] + out[n];
        if (scale_and_bias)
            for (n = 0; n < 32; n++)
                out[n] = out[n] * scale + bias;
        out += 32;
    }
    memcpy(vstate, vb, 576 * sizeof(float));
}

/* ------------------------------------------------------------------ */
/* a13 autocorrelate / inverse filter / chirp, aacsbr.c:1232-1334       */
/* ------------------------------------------------------------------ */
static void autocorrelate(const float x[40][2], float phi[3][2][2], int lag)
{
    int i;
    float real_sum = 0.0f, imag_sum = 0.0f;
    if (lag) {
        for (i = 1; i < 38; i++) {
            real_sum += x[i][0] * x[i + lag][0] + x[i][1] * x[i + lag][1];
            imag_sum += x[i][0] * x[i + lag][1] - x[i][1] * x[i + lag][0];
        }
        phi[2 - lag][1][0] = real_sum + x[0][0] * x[lag][0] + x[0][1] * x[lag][1];
        phi[2 - lag][1][1] = imag_sum + x[0][0] * x[lag][1] - x[0][1] * x[lag][0];
        if (lag == 1) {
            phi[0][0][0] = real_sum + x[38][0] * x[39][0] + x[38][1] * x[39][1];
            phi[0][0][1] = imag_sum + x[38][0] * x[39][1] - x[38][1] * x[39][0];
        }
    } else {
        for (i = 1; i < 38; i++)
            real_sum += x[i][0] * x[i][0] + x[i][1] * x[i][1];
        phi[2][1][0] = real_sum + x[0][0] * x[0][0] + x[0][1] * x[0][1];
        phi[1][0][0] = real_sum + x[38][0] * x[38][0] + x[38][1] * x[38][1];
    }
}

static void hf_inverse_filter(float (*alpha0)[2], float (*alpha1)[2],
                              const float X_low[32][40][2], int k0)
{
    int k;
    for (k = 0; k < k0; k++) {
        float phi[3][2][2], dk;
        autocorrelate(X_low[k], phi, 0);
        autocorrelate(X_low[k], phi, 1);
        autocorrelate(X_low[k], phi, 2);

        dk = phi[2][1][0] * phi[1][0][0] -
             (phi[1][1][0] * phi[1][1][0] + phi[1][1][1] * phi[1][1][1]) / 1.000001f;

        if (!dk) {
            alpha1[k][0] = 0;
            alpha1[k][1] = 0;
        } else {
            float temp_real, temp_im;
            temp_real = phi[0][0][0] * phi[1][1][0] -
                        phi[0][0][1] * phi[1][1][1] -
                        phi[0][1][0] * phi[1][0][0];
            temp_im   = phi[0][0][0] * phi[1][1][1] +
                        phi[0][0][1] * phi[1][1][0] -
                        phi[0][1][1] * phi[1][0][0];
            alpha1[k][0] = temp_real / dk;
            alpha1[k][1] = temp_im   / dk;
        }

        if (!phi[1][0][0]) {
            alpha0[k][0] = 0;
            alpha0[k][1] = 0;
        } else {
            float temp_real, temp_im;
            temp_real = phi[0][0][0] + alpha1[k][0] * phi[1][1][0] +
                                       alpha1[k][1] * phi[1][1][1];
            temp_im   = phi[0][0][1] + alpha1[k][1] * phi[1][1][0] -
                                       alpha1[k][0] * phi[1][1][1];
            alpha0[k][0] = -temp_real / phi[1][0][0];
            alpha0[k][1] = -temp_im   / phi[1][0][0];
        }

        if (alpha1[k][0] * alpha1[k][0] + alpha1[k][1] * alpha1[k][1] >= 16.0f ||
            alpha0[k][0] * alpha0[k][0] + alpha0[k][1] * alpha0[k][1] >= 16.0f) {
            alpha1[k][0] = 0;
            alpha1[k][1] = 0;
            alpha0[k][0] = 0;
            alpha0[k][1] = 0;
        }
    }
}

static void chirp(const HeaacSbrHeader *h, const HeaacSbrChannel *c, sbr_ch *d)
{
    static const float bw_tab[] = { 0.0f, 0.75f, 0.9f, 0.98f };
    int i;
    for (i = 0; i < h->n_q; i++) {
        float new_bw;
        if (c->bs_invf_mode[0][i] + c->bs_invf_mode[1][i] == 1)
            new_bw = 0.6f;
        else
            new_bw = bw_tab[c->bs_invf_mode[0][i]];
        if (new_bw < d->bw_array[i])
            new_bw = 0.75f    * new_bw + 0.25f    * d->bw_array[i];
        else
            new_bw = 0.90625f * new_bw + 0.09375f * d->bw_array[i];
        d->bw_array[i] = new_bw < 0.015625f ? 0.0f : new_bw;
    }
}

/* a12 sbr_lf_gen, aacsbr.c:1337-1357 */
static void lf_gen(float X_low[32][40][2], const float W[2][32][32][2], int kx_new, int kx_old)
{
    int i, k;
    memset(X_low, 0, 32 * sizeof(*X_low));
    for (k = 0; k < kx_new; k++)
        for (i = 8; i < 40; i++) {
            X_low[k][i][0] = W[1][i - 8][k][0];
            X_low[k][i][1] = W[1][i - 8][k][1];
        }
    for (k = 0; k < kx_old; k++)
        for (i = 0; i < 8; i++) {
            X_low[k][i][0] = W[0][i + 24][k][0];
            X_low[k][i][1] = W[0][i + 24][k][1];
        }
}

/* a14 sbr_hf_gen, aacsbr.c:1360-1409 */
static int hf_gen(const HeaacSbrHeader *h, float X_high[64][40][2], const float X_low[32][40][2],
                  const float (*alpha0)[2], const float (*alpha1)[2],
                  const float bw_array[5], const uint8_t *t_env, int bs_num_env)
{
    int i, j, x, g = 0, k = h->kx;
    for (j = 0; j < h->num_patches; j++) {
        for (x = 0; x < h->patch_num_subbands[j]; x++, k++) {
            float alpha[4];
            const int p = h->patch_start_subband[j] + x;
            while (g <= h->n_q && k >= h->f_tablenoise[g])
                g++;
            g--;
            if (g < 0)
                return -1;
            alpha[0] = alpha1[p][0] * bw_array[g] * bw_array[g];
            alpha[1] = alpha1[p][1] * bw_array[g] * bw_array[g];
            alpha[2] = alpha0[p][0] * bw_array[g];
            alpha[3] = alpha0[p][1] * bw_array[g];
            for (i = 2 * t_env[0]; i < 2 * t_env[bs_num_env]; i++) {
                const int idx = i + ENV_ADJ_OFFSET;
                X_high[k][idx][0] =
                    X_low[p][idx - 2][0] * alpha[0] -
                    X_low[p][idx - 2][1] * alpha[1] +
                    X_low[p][idx - 1][0] * alpha[2] -
                    X_low[p][idx - 1][1] * alpha[3] +
                    X_low[p][idx][0];
                X_high[k][idx][1] =
                    X_low[p][idx - 2][1] * alpha[0] +
                    X_low[p][idx - 2][0] * alpha[1] +
                    X_low[p][idx - 1][1] * alpha[2] +
                    X_low[p][idx - 1][0] * alpha[3] +
                    X_low[p][idx][1];
            }
        }
    }
    if (k < h->m + h->kx)
        memset(X_high + k, 0, (h->m + h->kx - k) * sizeof(*X_high));
    return 0;
}

/* a19 sbr_x_gen, aacsbr.c:1412-1446 */
static void x_gen(float X[2][38][64], const float X_low[32][40][2], const float Y[2][38][64][2],
                  int kx0, int m0, int kx1, int m1, int t_env_num_env_old)
{
    int k, i;
    const int i_f = 32;
    const int i_Temp = FFMAX(2 * t_env_num_env_old - i_f, 0);
    memset(X, 0, 2 * sizeof(*X));
    for (k = 0; k < kx0; k++)
        for (i = 0; i < i_Temp; i++) {
            X[0][i][k] = X_low[k][i + ENV_ADJ_OFFSET][0];
            X[1][i][k] = X_low[k][i + ENV_ADJ_OFFSET][1];
        }
    for (; k < kx0 + m0; k++)
        for (i = 0; i < i_Temp; i++) {
            X[0][i][k] = Y[0][i + i_f][k][0];
            X[1][i][k] = Y[0][i + i_f][k][1];
        }
    for (k = 0; k < kx1; k++)
        for (i = i_Temp; i < 38; i++) {
            X[0][i][k] = X_low[k][i + ENV_ADJ_OFFSET][0];
            X[1][i][k] = X_low[k][i + ENV_ADJ_OFFSET][1];
        }
    for (; k < kx1 + m1; k++)
        for (i = i_Temp; i < i_f; i++) {
            X[0][i][k] = Y[1][i][k][0];
            X[1][i][k] = Y[1][i][k][1];
        }
}

/* a15 sbr_mapping, aacsbr.c:1451-1496 */
static void mapping(const HeaacSbrHeader *h, const HeaacSbrChannel *c, sbr_ch *d, sbr_scratch *s)
{
    int e, i, m;
    const int kx1 = h->kx;
    memset(d->s_indexmapped[1], 0, 7 * sizeof(d->s_indexmapped[1]));
    for (e = 0; e < c->bs_num_env; e++) {
        const unsigned int ilim = h->n[c->bs_freq_res[e + 1]];
        const uint8_t *table = c->bs_freq_res[e + 1] ? h->f_tablehigh : h->f_tablelow;
        int k;
        for (i = 0; i < (int)ilim; i++)
            for (m = table[i]; m < table[i + 1]; m++)
                s->e_origmapped[e][m - kx1] = d->env_facs[e + 1][i];

        k = (c->bs_num_noise > 1) && (c->t_env[e] >= c->t_q[1]);
        for (i = 0; i < h->n_q; i++)
            for (m = h->f_tablenoise[i]; m < h->f_tablenoise[i + 1]; m++)
                s->q_mapped[e][m - kx1] = d->noise_facs[k + 1][i];

        for (i = 0; i < h->n[1]; i++) {
            if (c->bs_add_harmonic_flag) {
                const unsigned int m_midpoint = (h->f_tablehigh[i] + h->f_tablehigh[i + 1]) >> 1;
                d->s_indexmapped[e + 1][m_midpoint - kx1] = c->bs_add_harmonic[i] *
                    (e >= c->e_a[1] || (d->s_indexmapped[0][m_midpoint - kx1] == 1));
            }
        }
        for (i = 0; i < (int)ilim; i++) {
            int additional_sinusoid_present = 0;
            for (m = table[i]; m < table[i + 1]; m++) {
                if (d->s_indexmapped[e + 1][m - kx1]) {
                    additional_sinusoid_present = 1;
                    break;
                }
            }
            memset(&s->s_mapped[e][table[i] - kx1], additional_sinusoid_present,
                   (table[i + 1] - table[i]) * sizeof(s->s_mapped[e][0]));
        }
    }
    memcpy(d->s_indexmapped[0], d->s_indexmapped[c->bs_num_env], sizeof(d->s_indexmapped[0]));
}

/* a16 sbr_env_estimate, aacsbr.c:1499-1546 */
static void env_estimate(const HeaacSbrHeader *h, const HeaacSbrChannel *c, sbr_scratch *s)
{
    int e, i, m;
    const int kx1 = h->kx;
    float (*X_high)[40][2] = s->X_high;
    if (h->bs_interpol_freq) {
        for (e = 0; e < c->bs_num_env; e++) {
            const float recip_env_size = 0.5f / (c->t_env[e + 1] - c->t_env[e]);
            int ilb = c->t_env[e]     * 2 + ENV_ADJ_OFFSET;
            int iub = c->t_env[e + 1] * 2 + ENV_ADJ_OFFSET;
            for (m = 0; m < h->m; m++) {
                float sum = 0.0f;
                for (i = ilb; i < iub; i++)
                    sum += X_high[m + kx1][i][0] * X_high[m + kx1][i][0] +
                           X_high[m + kx1][i][1] * X_high[m + kx1][i][1];
                s->e_curr[e][m] = sum * recip_env_size;
            }
        }
    } else {
        int k, p;
        for (e = 0; e < c->bs_num_env; e++) {
            const int env_size = 2 * (c->t_env[e + 1] - c->t_env[e]);
            int ilb = c->t_env[e]     * 2 + ENV_ADJ_OFFSET;
            int iub = c->t_env[e + 1] * 2 + ENV_ADJ_OFFSET;
            const uint8_t *table = c->bs_freq_res[e + 1] ? h->f_tablehigh : h->f_tablelow;
            for (p = 0; p < h->n[c->bs_freq_res[e + 1]]; p++) {
                float sum = 0.0f;
                const int den = env_size * (table[p + 1] - table[p]);
                for (k = table[p]; k < table[p + 1]; k++)
                    for (i = ilb; i < iub; i++)
                        sum += X_high[k][i][0] * X_high[k][i][0] +
                               X_high[k][i][1] * X_high[k][i][1];
                sum /= den;
                for (k = table[p]; k < table[p + 1]; k++)
                    s->e_curr[e][k - kx1] = sum;
            }
        }
    }
}

/* a17 sbr_gain_calc, aacsbr.c:1552-1605 */
static void gain_calc(const HeaacSbrHeader *h, const HeaacSbrChannel *c, sbr_ch *d, sbr_scratch *s)
{
    static const float limgain[4] = { 0.70795, 1.0, 1.41254, 10000000000 };
    const int kx1 = h->kx;
    int e, k, m;
    for (e = 0; e < c->bs_num_env; e++) {
        int delta = !((e == c->e_a[1]) || (e == c->e_a[0]));
        for (k = 0; k < h->n_lim; k++) {
            float gain_boost, gain_max;
            float sum[2] = { 0.0f, 0.0f };
            const int m0 = h->f_tablelim[k] - kx1, m1 = h->f_tablelim[k + 1] - kx1;
            for (m = m0; m < m1; m++) {
                const float temp = s->e_origmapped[e][m] / (1.0f + s->q_mapped[e][m]);
                s->q_m[e][m] = sqrtf(temp * s->q_mapped[e][m]);
                s->s_m[e][m] = sqrtf(temp * d->s_indexmapped[e + 1][m]);
                if (!s->s_mapped[e][m]) {
                    s->gain[e][m] = sqrtf(s->e_origmapped[e][m] /
                                          ((1.0f + s->e_curr[e][m]) *
                                           (1.0f + s->q_mapped[e][m] * delta)));
                } else {
                    s->gain[e][m] = sqrtf(s->e_origmapped[e][m] * s->q_mapped[e][m] /
                                          ((1.0f + s->e_curr[e][m]) *
                                           (1.0f + s->q_mapped[e][m])));
                }
            }
            for (m = m0; m < m1; m++) {
                sum[0] += s->e_origmapped[e][m];
                sum[1] += s->e_curr[e][m];
            }
            gain_max = limgain[h->bs_limiter_gains] * sqrtf((FLT_EPSILON + sum[0]) / (FLT_EPSILON + sum[1]));
            gain_max = FFMIN(100000, gain_max);
            for (m = m0; m < m1; m++) {
                float q_m_max  = s->q_m[e][m] * gain_max / s->gain[e][m];
                s->q_m[e][m]   = FFMIN(s->q_m[e][m], q_m_max);
                s->gain[e][m]  = FFMIN(s->gain[e][m], gain_max);
            }
            sum[0] = sum[1] = 0.0f;
            for (m = m0; m < m1; m++) {
                sum[0] += s->e_origmapped[e][m];
                sum[1] += s->e_curr[e][m] * s->gain[e][m] * s->gain[e][m]
                          + s->s_m[e][m] * s->s_m[e][m]
                          + (delta && !s->s_m[e][m]) * s->q_m[e][m] * s->q_m[e][m];
            }
            gain_boost = sqrtf((FLT_EPSILON + sum[0]) / (FLT_EPSILON + sum[1]));
            gain_boost = FFMIN(1.584893192, gain_boost);
            for (m = m0; m < m1; m++) {
                s->gain[e][m] *= gain_boost;
                s->q_m[e][m]  *= gain_boost;
                s->s_m[e][m]  *= gain_boost;
            }
        }
    }
}

/* a18 sbr_hf_assemble, aacsbr.c:1608-1714 */
static void hf_assemble(const HeaacSbrHeader *h, const HeaacSbrFrame *fr, const HeaacSbrChannel *c,
                        sbr_ch *d, sbr_scratch *s)
{
    const or_tables *t = oracle_tables();
    int e, i, j, m;
    const int h_SL = 4 * !h->bs_smoothing_mode;
    const int kx = h->kx;
    const int m_max = h->m;
    static const float h_smooth[5] = {
        0.33333333333333, 0.30150283239582, 0.21816949906249,
        0.11516383427084, 0.03183050093751,
    };
    static const int8_t phi[2][4] = { { 1, 0, -1, 0 }, { 0, 1, 0, -1 } };
    float (*g_temp)[48] = d->g_temp, (*q_temp)[48] = d->q_temp;
    float (*Y)[38][64][2] = d->Y;
    float (*X_high)[40][2] = s->X_high;
    int indexnoise = d->f_indexnoise;
    int indexsine  = d->f_indexsine;
    const int *e_a_dummy = NULL; (void)e_a_dummy;
    memcpy(Y[0], Y[1], sizeof(Y[0]));

    if (fr->reset) {
        for (i = 0; i < h_SL; i++) {
            memcpy(g_temp[i + 2 * c->t_env[0]], s->gain[0], m_max * sizeof(s->gain[0][0]));
            memcpy(q_temp[i + 2 * c->t_env[0]], s->q_m[0],  m_max * sizeof(s->q_m[0][0]));
        }
    } else if (h_SL) {
        memcpy(g_temp[2 * c->t_env[0]], g_temp[2 * c->t_env_num_env_old], 4 * sizeof(g_temp[0]));
        memcpy(q_temp[2 * c->t_env[0]], q_temp[2 * c->t_env_num_env_old], 4 * sizeof(q_temp[0]));
    }

    for (e = 0; e < c->bs_num_env; e++)
        for (i = 2 * c->t_env[e]; i < 2 * c->t_env[e + 1]; i++) {
            memcpy(g_temp[h_SL + i], s->gain[e], m_max * sizeof(s->gain[0][0]));
            memcpy(q_temp[h_SL + i], s->q_m[e],  m_max * sizeof(s->q_m[0][0]));
        }

    for (e = 0; e < c->bs_num_env; e++) {
        for (i = 2 * c->t_env[e]; i < 2 * c->t_env[e + 1]; i++) {
            int phi_sign = (1 - 2 * (kx & 1));

            if (h_SL && e != c->e_a[0] && e != c->e_a[1]) {
                for (m = 0; m < m_max; m++) {
                    const int idx1 = i + h_SL;
                    float g_filt = 0.0f;
                    for (j = 0; j <= h_SL; j++)
                        g_filt += g_temp[idx1 - j][m] * h_smooth[j];
                    Y[1][i][m + kx][0] = X_high[m + kx][i + ENV_ADJ_OFFSET][0] * g_filt;
                    Y[1][i][m + kx][1] = X_high[m + kx][i + ENV_ADJ_OFFSET][1] * g_filt;
                }
            } else {
                for (m = 0; m < m_max; m++) {
                    const float g_filt = g_temp[i + h_SL][m];
                    Y[1][i][m + kx][0] = X_high[m + kx][i + ENV_ADJ_OFFSET][0] * g_filt;
                    Y[1][i][m + kx][1] = X_high[m + kx][i + ENV_ADJ_OFFSET][1] * g_filt;
                }
            }

            if (e != c->e_a[0] && e != c->e_a[1]) {
                for (m = 0; m < m_max; m++) {
                    indexnoise = (indexnoise + 1) & 0x1ff;
                    if (s->s_m[e][m]) {
                        Y[1][i][m + kx][0] += s->s_m[e][m] * phi[0][indexsine];
                        Y[1][i][m + kx][1] += s->s_m[e][m] * (phi[1][indexsine] * phi_sign);
                    } else {
                        float q_filt;
                        if (h_SL) {
                            const int idx1 = i + h_SL;
                            q_filt = 0.0f;
                            for (j = 0; j <= h_SL; j++)
                                q_filt += q_temp[idx1 - j][m] * h_smooth[j];
                        } else {
                            q_filt = q_temp[i][m];
                        }
                        Y[1][i][m + kx][0] += q_filt * t->noise[indexnoise][0];
                        Y[1][i][m + kx][1] += q_filt * t->noise[indexnoise][1];
                    }
                    phi_sign = -phi_sign;
                }
            } else {
                indexnoise = (indexnoise + m_max) & 0x1ff;
                for (m = 0; m < m_max; m++) {
                    Y[1][i][m + kx][0] += s->s_m[e][m] * phi[0][indexsine];
                    Y[1][i][m + kx][1] += s->s_m[e][m] * (phi[1][indexsine] * phi_sign);
                    phi_sign = -phi_sign;
                }
            }
            indexsine = (indexsine + 1) & 3;
        }
    }
    d->f_indexnoise = indexnoise;
    d->f_indexsine  = indexsine;
}

/* ------------------------------------------------------------------ */
/* state record <-> working set                                         */
/* ------------------------------------------------------------------ */
static void sbr_unpack(sbr_ch *d, const float *st, const HeaacSbrChannel *c)
{
    int j;
    uint32_t u;
    memset(d, 0, sizeof(*d));
    memcpy(d->xbuf + 1024, st + HEAAC_SBR_XHIST, 288 * sizeof(float));
    memcpy(d->W[1][24], st + HEAAC_SBR_WTAIL, 8 * 32 * 2 * sizeof(float));
    memcpy(d->Y[1][32], st + HEAAC_SBR_YTAIL, 6 * 64 * 2 * sizeof(float));
    for (j = 0; j < 4; j++) {
        memcpy(d->g_temp[2 * c->t_env_num_env_old + j], st + HEAAC_SBR_GTAIL + 48 * j, 48 * sizeof(float));
        memcpy(d->q_temp[2 * c->t_env_num_env_old + j], st + HEAAC_SBR_QTAIL + 48 * j, 48 * sizeof(float));
    }
    memcpy(d->bw_array, st + HEAAC_SBR_BW, 5 * sizeof(float));
    memcpy(&u, st + HEAAC_SBR_IDXNOISE, 4); d->f_indexnoise = u;
    memcpy(&u, st + HEAAC_SBR_IDXSINE, 4);  d->f_indexsine = u;
    memcpy(d->s_indexmapped[0], st + HEAAC_SBR_SIDX, 48);
}

static void sbr_pack(float *st, const float *st_in, const sbr_ch *d, const HeaacSbrChannel *c,
                     const HeaacSbrHeader *h, int started)
{
    int j;
    uint32_t u;
    if (st != st_in)
        memcpy(st, st_in, HEAAC_ST_SBR * sizeof(float));
    memcpy(st + HEAAC_SBR_XHIST, d->xbuf + 1024, 288 * sizeof(float));
    memcpy(st + HEAAC_SBR_WTAIL, d->W[1][24], 8 * 32 * 2 * sizeof(float));
    if (!started)
        return;
    memcpy(st + HEAAC_SBR_YTAIL, d->Y[1][32], 6 * 64 * 2 * sizeof(float));
    if (!h->bs_smoothing_mode) {
        /* Only the m[1] bands of the current header are live history: a wider range arrives only
         * with sbr->reset, which refills the rows from gain[0] / q_m[0] (aacsbr.c:1632-1637) before
         * anything reads them.  The reference's persistent g_temp rows keep whatever an older, wider
         * header left above m[1]; the state record defines those words as zero. */
        const int m_max = h->m < 48 ? h->m : 48;
        for (j = 0; j < 4; j++) {
            memset(st + HEAAC_SBR_GTAIL + 48 * j, 0, 48 * sizeof(float));
            memset(st + HEAAC_SBR_QTAIL + 48 * j, 0, 48 * sizeof(float));
            memcpy(st + HEAAC_SBR_GTAIL + 48 * j, d->g_temp[2 * c->t_env[c->bs_num_env] + j], m_max * sizeof(float));
            memcpy(st + HEAAC_SBR_QTAIL + 48 * j, d->q_temp[2 * c->t_env[c->bs_num_env] + j], m_max * sizeof(float));
        }
    }
    memcpy(st + HEAAC_SBR_BW, d->bw_array, 5 * sizeof(float));
    u = d->f_indexnoise; memcpy(st + HEAAC_SBR_IDXNOISE, &u, 4);
    u = d->f_indexsine;  memcpy(st + HEAAC_SBR_IDXSINE, &u, 4);
    memcpy(st + HEAAC_SBR_SIDX, d->s_indexmapped[0], 48);
}

/* ------------------------------------------------------------------ */
/* a21 ff_sbr_apply, aacsbr.c:1716-1771 + a8 with bias 0, one frame      */
/* ------------------------------------------------------------------ */
typedef struct {
    float *W, *Xlow, *Xhigh, *Y, *Xsbr, *X;
} dump_ptrs;

static int he_frame(int cfg, int flags, int simd, const float *coeffs, const HeaacIcs *ics,
                    const HeaacSbrFrame *fr, const HeaacSbrHeader *hdr_tab, size_t n_hdr,
                    const HeaacPsFrame *ps,
                    const float *st_in, float *st_out, float *ret[2], const dump_ptrs *dp)
{
    const int cpe = (cfg == HEAAC_CFG_HEV1);
    const int ncore = cpe ? 2 : 1;
    const int with_ps = (cfg == HEAAC_CFG_HEV2);
    const HeaacSbrHeader *h;
    static __thread sbr_ch d[2];                  /* (thread-local: the CPU baseline runs the oracle on all cores) */
    static __thread sbr_scratch s;
    int ch, nch;
    /* state record sub-offsets */
    size_t off_saved[2], off_sbr[2], off_syn[2], off_ps = 0;
    /* ac->sf_scale, ac->add_bias: the C conversion's or the SIMD configuration's (aacdec.c:573-581) */
    const float sf_scale = simd ? -1.0f / 1024.0f : HEAAC_SF_SCALE;
    const float add_bias = simd ? 0.0f : HEAAC_ADD_BIAS;

    if (n_hdr && fr->hdr >= n_hdr)
        return HEAAC_ERR_ARG;
    h = &hdr_tab[fr->hdr];
    /* The reference keeps these arrays in the per-stream context (av_mallocz'ed,
     * sbr.h:140-177).  Entries no stage writes this frame -- e.g. gain[e][m] for
     * bands above the last limiter border when the last patch was dropped
     * (aacsbr.c:538-539) -- are therefore zero in a fresh context; the oracle
     * and the HIP path define them as zero. */
    memset(&s, 0, sizeof(s));

    if (cpe) {
        off_saved[0] = 0; off_saved[1] = 512;
        off_sbr[0] = 1024; off_sbr[1] = 1024 + HEAAC_ST_SBR;
        off_syn[0] = 1024 + 2 * HEAAC_ST_SBR; off_syn[1] = off_syn[0] + HEAAC_ST_SYNTH;
    } else {
        off_saved[0] = 0; off_sbr[0] = 512;
        off_syn[0] = 512 + HEAAC_ST_SBR; off_syn[1] = off_syn[0] + HEAAC_ST_SYNTH;
        off_ps = off_syn[1] + HEAAC_ST_SYNTH;
        off_saved[1] = off_sbr[1] = 0;
    }

    /* imdct_and_windowing with imdct_bias = 0 (aacdec.c:1906) */
    for (ch = 0; ch < ncore; ch++) {
        float saved[512];
        memcpy(saved, st_in + off_saved[ch], sizeof(saved));
        oracle_imdct_and_windowing(coeffs + 1024 * ch, &ics[ch], saved, ret[ch], 0.0f);
        memcpy(st_out + off_saved[ch], saved, sizeof(saved));
    }

    for (ch = 0; ch < ncore; ch++) {
        sbr_unpack(&d[ch], st_in + off_sbr[ch], &fr->ch[ch]);
        if (fr->reset)
            d[ch].f_indexnoise = 0;      /* sbr_make_f_derived, aacsbr.c:587-588 */
    }

    if (fr->start)
        dequant(fr, h, cpe, &d[0], &d[1]);

    for (ch = 0; ch < ncore; ch++) {
        const HeaacSbrChannel *c = &fr->ch[ch];
        /* kx[1]/m[1] before any header was seen are 32/0 (aacsbr.c:130) */
        const int kx1 = h->kx, m1 = h->m;
        qmf_analysis(ret[ch], d[ch].xbuf, d[ch].W, 1 / (-1024 * sf_scale));
        if (dp && dp->W)
            memcpy(dp->W + ch * 32 * 32 * 2, d[ch].W[1], sizeof(d[ch].W[1]));
        lf_gen(s.X_low, d[ch].W, kx1, fr->kx_old);
        if (fr->start) {
            int e_a[2] = { c->e_a[0], c->e_a[1] };
            (void)e_a;
            hf_inverse_filter(s.alpha0, s.alpha1, s.X_low, h->k0);
            chirp(h, c, &d[ch]);
            if (hf_gen(h, s.X_high, s.X_low, s.alpha0, s.alpha1, d[ch].bw_array, c->t_env, c->bs_num_env) < 0)
                return HEAAC_ERR_ARG;
            mapping(h, c, &d[ch], &s);
            env_estimate(h, c, &s);
            gain_calc(h, c, &d[ch], &s);
            hf_assemble(h, fr, c, &d[ch], &s);
        }
        if (dp && ch == 0) {
            if (dp->Xlow)  memcpy(dp->Xlow, s.X_low, sizeof(s.X_low));
            if (dp->Xhigh) memcpy(dp->Xhigh, s.X_high, sizeof(s.X_high));
            if (dp->Y)     memcpy(dp->Y, d[ch].Y[1], sizeof(d[ch].Y[1]));
        }
        if (!fr->start) {
            /* No HF stage this frame, so no Y[0] <- Y[1] either (aacsbr.c:1629): the reference's
             * sbr_x_gen would read a Y[0] that is one frame older than the state record holds, and,
             * above kx[1], the whole stale Y[1] of the last frame that had one.  Neither is live state
             * of an error-free stream (start = 0 mid-stream follows a payload that failed to parse);
             * the record's rule: the first i_Temp slots take the Y tail it carries,
             * Y[1][i < 32] reads as zero. */
            memcpy(d[ch].Y[0], d[ch].Y[1], sizeof(d[ch].Y[0]));
        }
        x_gen(s.X[ch], s.X_low, d[ch].Y, fr->kx_old, fr->m_old, kx1, m1, c->t_env_num_env_old);
        sbr_pack(st_out + off_sbr[ch], st_in + off_sbr[ch], &d[ch], c, h, fr->start);
    }
    if (dp && dp->Xsbr)
        memcpy(dp->Xsbr, s.X, sizeof(s.X));

    nch = ncore;
    if (with_ps) {
        if (st_out != st_in)
            memcpy(st_out + off_ps, st_in + off_ps, HEAAC_ST_PS * sizeof(float));
        if (ps->start)
            or_ps_apply(ps, st_out + off_ps, s.X[0], s.X[1], h->kx + h->m);
        else
            memcpy(s.X[1], s.X[0], sizeof(s.X[0]));
        nch = 2;
    }
    if (dp && dp->X)
        memcpy(dp->X, s.X, sizeof(s.X));

    for (ch = 0; ch < nch; ch++) {
        float v[1152];
        memcpy(v, st_in + off_syn[ch], sizeof(v));
        if (flags & HEAAC_HE_DOWNSAMPLED) {
            /* downsampled = ext_sample_rate < sbr->sample_rate (aacsbr.c:1719): div = 1, 1024 samples */
            static __thread float Xds[2][32][64];
            int i;
            for (i = 0; i < 32; i++) {
                memcpy(Xds[0][i], s.X[ch][0][i], sizeof(Xds[0][i]));
                memcpy(Xds[1][i], s.X[ch][1][i], sizeof(Xds[1][i]));
            }
            oracle_qmf_synthesis_ds(&Xds[0][0][0], v, ret[ch], -1024 * sf_scale, add_bias);
        } else {
            qmf_synthesis(ret[ch], s.X[ch], v, add_bias, -1024 * sf_scale);
        }
        memcpy(st_out + off_syn[ch], v, sizeof(v));
    }
    return 0;
}

static int cfg_words(int cfg)
{
    switch (cfg) {
    case HEAAC_CFG_HEV1:      return HEAAC_STATE_WORDS_HEV1;
    case HEAAC_CFG_HEV1_MONO: return HEAAC_STATE_WORDS_HEV1_MONO;
    case HEAAC_CFG_HEV2:      return HEAAC_STATE_WORDS_HEV2;
    }
    return -1;
}

int oracle_he_decode_batch(int cfg, const float *coeffs, const HeaacIcs *ics,
                           const HeaacSbrFrame *sbr, const HeaacSbrHeader *hdr, size_t n_hdr,
                           const HeaacPsFrame *ps,
                           const float *state_in, float *state_out,
                           void *pcm, int pcm_format, size_t n)
{
    return oracle_he_decode_batch_ex(cfg, 0, coeffs, ics, sbr, hdr, n_hdr, ps, state_in, state_out, pcm, pcm_format, n);
}

int oracle_he_decode_batch_ex(int cfg, int flags, const float *coeffs, const HeaacIcs *ics,
                              const HeaacSbrFrame *sbr, const HeaacSbrHeader *hdr, size_t n_hdr,
                              const HeaacPsFrame *ps,
                              const float *state_in, float *state_out,
                              void *pcm, int pcm_format, size_t n)
{
    const int words = cfg_words(cfg);
    const int ncore = (cfg == HEAAC_CFG_HEV1) ? 2 : 1;
    const int nout = (cfg == HEAAC_CFG_HEV1_MONO) ? 1 : 2;
    size_t f;
    if (words < 0)
        return HEAAC_ERR_ARG;
    oracle_tables();
    for (f = 0; f < n; f++) {
        float retbuf[2][2048];
        float *ret[2] = { retbuf[0], retbuf[1] };
        float *tmp = NULL;
        const float *sin_ = state_in + f * words;
        float *sout = state_out + f * words;
        int r;
        if (sin_ == sout) {             /* in-place: work from a copy */
            tmp = malloc(words * sizeof(float));
            memcpy(tmp, sin_, words * sizeof(float));
            sin_ = tmp;
        }
        r = he_frame(cfg, flags, pcm_format == HEAAC_PCM_S16_INTERLEAVED_SSE2, coeffs + f * ncore * 1024, ics + f * ncore, &sbr[f], hdr, n_hdr,
                     ps ? &ps[f] : NULL, sin_, sout, ret, NULL);
        free(tmp);
        if (r < 0)
            return r;
        or_store_pcm(pcm, pcm_format, f, nout, (flags & HEAAC_HE_DOWNSAMPLED) ? 1024 : 2048, ret);
    }
    return 0;
}

int oracle_he_decode_debug(int cfg, const float *coeffs, const HeaacIcs *ics,
                           const HeaacSbrFrame *sbr, const HeaacSbrHeader *hdr,
                           const HeaacPsFrame *ps,
                           const float *state_in, float *state_out,
                           float *pcm_f32, float *dump_W, float *dump_Xlow,
                           float *dump_Xhigh, float *dump_Y, float *dump_Xsbr,
                           float *dump_X)
{
    dump_ptrs dp = { dump_W, dump_Xlow, dump_Xhigh, dump_Y, dump_Xsbr, dump_X };
    const int nout = (cfg == HEAAC_CFG_HEV1_MONO) ? 1 : 2;
    float retbuf[2][2048];
    float *ret[2] = { retbuf[0], retbuf[1] };
    int r;
    if (cfg_words(cfg) < 0)
        return HEAAC_ERR_ARG;
    oracle_tables();
    r = he_frame(cfg, 0, 0, coeffs, ics, sbr, hdr, 0, ps, state_in, state_out, ret, &dp);
    if (r < 0)
        return r;
    if (pcm_f32)
        or_store_pcm(pcm_f32, HEAAC_PCM_F32_PLANAR, 0, nout, 2048, ret);
    return 0;
}
