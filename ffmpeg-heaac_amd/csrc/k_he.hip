// k_he.hip -- HE-AAC (SBR + Parametric Stereo) batched kernels for gfx950.
//
//   k_core_ana  : imdct_and_windowing (bias 0) + sbr_qmf_analysis      (a8, a11)
//   k_hfadj     : lf_gen, inverse filter, chirp, hf_gen, mapping,
//                 env_estimate, gain_calc, hf_assemble, x_gen          (a10, a12-a19)
//   k_ps        : hybrid analysis, decorrelation, stereo_processing,
//                 hybrid synthesis                                     (a22-a26)
//   k_synth     : sbr_qmf_synthesis + float_to_int16_interleave        (a20, a27)
//
// One wavefront owns one unit (an SBR channel, a PS frame, an output channel);
// workgroups are persistent and keep the immutable tables in LDS.  Stages hand
// W[32][32][2] and X[2][38][64] to each other through a workspace that the
// host sizes to stay inside the 256 MiB Infinity Cache (frames are processed in
// chunks), so the intermediates do not travel to HBM.
//
// Reference line numbers are libavcodec/aacsbr.c and aacps.c.
#include "k_core.h"
#include "kernels.h"

#define ENV_ADJ 2          // ENVELOPE_ADJUSTMENT_OFFSET, aacsbr.c:39

// exp2f(twice / 2.0f) for integer `twice`: exact powers of two, or sqrt(2)
// (0x3FB504F3, what glibc's exp2f(0.5f) returns) times a power of two.
// sbr_dequant's arguments are always multiples of 0.5 (aacsbr.c:1099-1125).
__device__ __forceinline__ float exp2_half(int twice)
{
    // outside the normal range (never reached by legal scalefactors): saturate
    // like exp2f does; the denormal side is rounded once more than libm's
    if (twice > 255) return __uint_as_float(0x7F800000u);
    if (twice < -252) return twice < -400 ? 0.0f : exp2_half(twice + 256) * 2.938735877055719e-39f; // 2^-128
    const int e = twice >> 1;
    const unsigned mant = (twice & 1) ? 0x3FB504F3u : 0x3F800000u;
    return __uint_as_float(mant + ((unsigned)e << 23));
}

// ===========================================================================
// K_A  core + QMF analysis
// ===========================================================================
#define ANA_WAVES 8
#define ANA_POOL  3456            // floats per wave: core 2560 | x 1312 + u 32*65

struct AnaLds {
    CoreLds core;
    float qmf_ds[320];
    float rot[64];                // SBR analysis MDCT: tcos[32], tsin[32]
    float pool[ANA_WAVES][ANA_POOL];
};

// sbr_qmf_analysis (aacsbr.c:1136-1169) on LDS data.
//   x    : 1312 floats, x[0..287] history, x[288..1311] = in * scale
//   u    : 32 rows of 65 floats scratch
//   g_W  : [32][32][2] output
__device__ __forceinline__ void qmf_analysis_wave(const float *qmf_ds, const float *rot,
                                                  const float *c16, const float *c32,
                                                  const float *x, float *u, float *g_W, int lane)
{
    // z[n] = ds[n] * x[319 - n]; f[k] = z[k]+z[k+64]+z[k+128]+z[k+192]+z[k+256]
    // lane = k keeps its 5 window taps in registers and walks the 32 slots.
    {
        const int k = lane;
        const float w0 = qmf_ds[k], w1 = qmf_ds[k + 64], w2 = qmf_ds[k + 128],
                    w3 = qmf_ds[k + 192], w4 = qmf_ds[k + 256];
        for (int i = 0; i < 32; i++) {
            const float *xs = x + 32 * i + 319 - k;
            const float f = w0 * xs[0] + w1 * xs[-64] + w2 * xs[-128] + w3 * xs[-192] + w4 * xs[-256];
            u[i * 65 + k] = f;
        }
    }
    wave_sync();
    // shuffle to the IMDCT input (:1155-1160): in[0] = f[0]; in[2k-1] = f[k];
    // in[2k] = -f[64-k] (k = 1..31); in[63] = f[32];  then ff_imdct_half (N = 128).
    if (lane < 32) {
        const float *f = u + lane * 65;
        float o[64];
        imdct128_reg([&](int j) -> float {
                         if (j == 0)  return f[0];
                         if (j == 63) return f[32];
                         return (j & 1) ? f[(j + 1) >> 1] : -f[64 - (j >> 1)];
                     }, o, rot, c16, c32);
        // W[1][i][k] = (-z[63-k], z[k])                                  (:1163-1166)
        float *row = u + lane * 65;       // own row: all of it is in registers now
#pragma unroll
        for (int k = 0; k < 32; k++) {
            row[2 * k]     = -o[63 - k];
            row[2 * k + 1] = o[k];
        }
    }
    wave_sync();
    // coalesced store: 2048 floats
    for (int t = lane; t < 2048; t += WAVE)
        g_W[t] = u[(t >> 6) * 65 + (t & 63)];
    wave_sync();
}

__global__ __launch_bounds__(ANA_WAVES * WAVE)
void k_core_ana(const float *__restrict__ g_tab, const uint16_t *__restrict__ g_rev,
                const float *__restrict__ g_coeffs, const HeaacIcs *__restrict__ g_ics,
                const float *g_state_in, float *g_state_out, int state_words,
                int ncore, int off_saved0, int off_sbr0,
                float *__restrict__ g_W, float scale, unsigned long long n_units)
{
    __shared__ AnaLds S;
    core_lds_init(S.core, g_tab, g_rev);
    for (int i = threadIdx.x; i < 320; i += blockDim.x) S.qmf_ds[i] = g_tab[TB_QMF_DS + i];
    for (int i = threadIdx.x; i < 64; i += blockDim.x)  S.rot[i] = g_tab[TB_ROT128A + i];
    __syncthreads();
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
    float *pool = S.pool[wave];
    const float *c16 = S.core.tab + TB_COS16, *c32 = S.core.tab + TB_COS32;

    for (unsigned long long u = (unsigned long long)blockIdx.x * ANA_WAVES + wave; u < n_units;
         u += (unsigned long long)gridDim.x * ANA_WAVES) {
        const unsigned long long f = u / ncore;
        const int ch = (int)(u - f * ncore);
        const float *st_in = g_state_in + f * state_words;
        float *st_out = g_state_out + f * state_words;
        const int off_saved = off_saved0 + ch * HEAAC_ST_SAVED;
        const int off_sbr = off_sbr0 + ch * HEAAC_ST_SBR;

        float *sbuf = pool, *zbuf = pool + 1024, *svd = pool + 2048;
        core_channel(S.core, g_coeffs + u * 1024, st_in + off_saved, st_out + off_saved,
                     g_ics[u], 0.0f, sbuf, zbuf, svd, lane);

        // After the core stage only sbuf (= out[1024], pool[0..1024)) is live.
        // x = [history 288 | in * scale 1024] goes to pool[2080 .. 3392); the
        // fold rows u[32][65] then overwrite pool[0 .. 2080).
        float *x = pool + 2080;
        const float *xh_in = st_in + off_sbr + HEAAC_SBR_XHIST;
        float *xh_out = st_out + off_sbr + HEAAC_SBR_XHIST;
        for (int i = lane; i < 288; i += WAVE) x[i] = xh_in[i];
        if (scale != 1.0f) {
            for (int i = lane; i < 1024; i += WAVE) x[288 + i] = sbuf[i] * scale;   // vector_fmul_scalar
        } else {
            for (int i = lane; i < 1024; i += WAVE) x[288 + i] = sbuf[i];
        }
        wave_sync();
        for (int i = lane; i < 288; i += WAVE) xh_out[i] = x[1024 + i];
        qmf_analysis_wave(S.qmf_ds, S.rot, c16, c32, x, pool + 0, g_W + u * 2048, lane);
    }
}

// ===========================================================================
// K_B  HF generation + envelope adjustment + x_gen, one wave per SBR channel
// ===========================================================================
#define HF_WAVES 6
#define XL_STRIDE 81              // X_low row: 40 slots * (re,im) + 1 pad (bank spread)
#define MAXM 48                   // e_origmapped[7][48] etc. in the reference (sbr.h:165-177)
#define MAXE 5

struct HfWave {
    float xlow[32 * XL_STRIDE];   // X_low[k][i][re,im]
    float alpha0[32][2], alpha1[32][2];
    float kc[MAXM][4];            // hf_gen coefficients alpha[0..3] per HF band m
    int   kp[MAXM];               // patch source band, -1: band above the last patch (zeros)
    float e_orig[MAXE][MAXM], q_map[MAXE][MAXM], e_curr[MAXE][MAXM];
    float gain[MAXE][MAXM], q_m[MAXE][MAXM], s_m[MAXE][MAXM];
    float ghist[4][MAXM], qhist[4][MAXM];
    float bw[8];
    int   env_of[40];
    uint8_t s_idx[MAXE + 1][MAXM];
    uint8_t s_map[MAXE][MAXM];
    HeaacSbrHeader h;
    HeaacSbrChannel c[2];
};

__device__ __forceinline__ void lds_copy_bytes(void *dst, const void *src, int bytes, int lane)
{
    // bytes % 4 == 0, both 4-byte aligned
    const uint32_t *s = reinterpret_cast<const uint32_t *>(src);
    uint32_t *d = reinterpret_cast<uint32_t *>(dst);
    for (int i = lane; i < bytes / 4; i += WAVE) d[i] = s[i];
}

// sbr_dequant (aacsbr.c:1089-1128) for one envelope scalefactor of channel ch.
__device__ __forceinline__ float deq_env(const HfWave &w, int coupling, int ch, int e, int i)
{
    if (coupling) {
        const int amp = w.c[0].bs_amp_res;
        const int q0 = w.c[0].env_facs_q[e][i], q1 = w.c[1].env_facs_q[e][i];
        // temp1 = exp2f(q0*alpha + 7), temp2 = exp2f((pan_offset - q1)*alpha)
        const float temp1 = exp2_half(amp ? 2 * q0 + 14 : q0 + 14);
        const float temp2 = exp2_half(amp ? 2 * (12 - q1) : 24 - q1);
        const float fac = temp1 / (1.0f + temp2);
        return ch ? fac * temp2 : fac;
    }
    const int amp = w.c[ch].bs_amp_res;
    const int q = w.c[ch].env_facs_q[e][i];
    return exp2_half((amp ? 2 * q : q) + 12);            // exp2f(alpha*q + 6)
}

__device__ __forceinline__ float deq_noise(const HfWave &w, int coupling, int ch, int e, int i)
{
    if (coupling) {
        const int q0 = w.c[0].noise_facs_q[e][i], q1 = w.c[1].noise_facs_q[e][i];
        const float temp1 = exp2_half(2 * (7 - q0));      // exp2f(NOISE_FLOOR_OFFSET - q0 + 1)
        const float temp2 = exp2_half(2 * (12 - q1));     // exp2f(12 - q1)
        const float fac = temp1 / (1.0f + temp2);
        return ch ? fac * temp2 : fac;
    }
    return exp2_half(2 * (6 - (int)w.c[ch].noise_facs_q[e][i]));   // exp2f(6 - q)
}

// X_high[kx + m][idx] of sbr_hf_gen (aacsbr.c:1388-1402), recomputed on demand.
__device__ __forceinline__ void xhigh(const HfWave &w, int m, int idx, float &re, float &im)
{
    const int p = w.kp[m];
    if (p < 0) { re = 0.0f; im = 0.0f; return; }
    const float *xl = w.xlow + p * XL_STRIDE + 2 * idx;
    const float a0 = w.kc[m][0], a1 = w.kc[m][1], a2 = w.kc[m][2], a3 = w.kc[m][3];
    re = xl[-4] * a0 - xl[-3] * a1 + xl[-2] * a2 - xl[-1] * a3 + xl[0];
    im = xl[-3] * a0 + xl[-4] * a1 + xl[-1] * a2 + xl[-2] * a3 + xl[1];
}

#define FFMIN_(a, b) ((a) > (b) ? (b) : (a))

__device__ __forceinline__ void hf_channel(HfWave &w, const float *__restrict__ g_noise,
                                           const HeaacSbrFrame *g_fr, const HeaacSbrHeader *g_hdr,
                                           int ch, const float *g_W,
                                           const float *st_in, float *st_out,
                                           float *g_X /* [2][38][64] */, int lane)
{
    // ---- parameters into LDS (uniform reads afterwards) ----
    const int hdr_idx = g_fr->hdr;
    lds_copy_bytes(&w.h, &g_hdr[hdr_idx], sizeof(HeaacSbrHeader), lane);
    lds_copy_bytes(&w.c[0], &g_fr->ch[0], 2 * sizeof(HeaacSbrChannel), lane);
    const int start = g_fr->start, reset = g_fr->reset;
    const int kx_old = g_fr->kx_old, m_old = g_fr->m_old;
    const int coupling = g_fr->bs_coupling;
    wave_sync();
    const HeaacSbrHeader &h = w.h;
    const HeaacSbrChannel &c = w.c[ch];
    const int kx = h.kx, m_max = h.m, n_q = h.n_q;
    const int num_env = c.bs_num_env;
    const int t0 = c.t_env[0], tL = c.t_env[num_env];
    const int h_SL = 4 * !h.bs_smoothing_mode;

    // ---- sbr_lf_gen (:1337-1357): W -> X_low, previous tail for slots 0..7 ----
    {
        const float2 *W2 = reinterpret_cast<const float2 *>(g_W);
        for (int t = lane; t < 1024; t += WAVE) {
            const int i = t >> 5, k = t & 31;
            float2 v = W2[t];
            if (k >= kx) v = make_float2(0.0f, 0.0f);
            float *d = w.xlow + k * XL_STRIDE + 2 * (i + 8);
            d[0] = v.x; d[1] = v.y;
        }
        const float2 *T2 = reinterpret_cast<const float2 *>(st_in + HEAAC_SBR_WTAIL);
        for (int t = lane; t < 256; t += WAVE) {
            const int i = t >> 5, k = t & 31;
            float2 v = T2[t];
            if (k >= kx_old) v = make_float2(0.0f, 0.0f);
            float *d = w.xlow + k * XL_STRIDE + 2 * i;
            d[0] = v.x; d[1] = v.y;
        }
        // new tail = W[1][24..31]
        float *To = st_out + HEAAC_SBR_WTAIL;
        for (int t = lane; t < 512; t += WAVE) To[t] = g_W[24 * 64 + t];
    }
    // small state words to LDS
    if (lane < 8) w.bw[lane] = lane < 5 ? st_in[HEAAC_SBR_BW + lane] : 0.0f;
    for (int t = lane; t < 4 * MAXM; t += WAVE) {
        (&w.ghist[0][0])[t] = st_in[HEAAC_SBR_GTAIL + t];
        (&w.qhist[0][0])[t] = st_in[HEAAC_SBR_QTAIL + t];
    }
    if (lane < 12)
        reinterpret_cast<uint32_t *>(&w.s_idx[0][0])[lane] =
            reinterpret_cast<const uint32_t *>(st_in + HEAAC_SBR_SIDX)[lane];
    unsigned idxnoise = __float_as_uint(st_in[HEAAC_SBR_IDXNOISE]);
    unsigned idxsine  = __float_as_uint(st_in[HEAAC_SBR_IDXSINE]);
    if (reset) idxnoise = 0;                     // sbr_make_f_derived, :587-588
    wave_sync();

    if (start) {
        // ---- sbr_hf_inverse_filter (:1261-1313) + autocorrelate (:1232-1255) ----
        if (lane < h.k0 && lane < 32) {
            const float *x = w.xlow + lane * XL_STRIDE;        // x[i][c] = x[2i + c]
            float r0 = 0.0f, r1 = 0.0f, i1 = 0.0f, r2 = 0.0f, i2 = 0.0f;
            for (int i = 1; i < 38; i++) {
                const float a = x[2 * i], b = x[2 * i + 1];
                r0 += a * a + b * b;
                r1 += a * x[2 * i + 2] + b * x[2 * i + 3];
                i1 += a * x[2 * i + 3] - b * x[2 * i + 2];
                r2 += a * x[2 * i + 4] + b * x[2 * i + 5];
                i2 += a * x[2 * i + 5] - b * x[2 * i + 4];
            }
            // lag 0: phi[2][1][0], phi[1][0][0]
            const float p210 = r0 + x[0] * x[0] + x[1] * x[1];
            const float p100 = r0 + x[76] * x[76] + x[77] * x[77];
            // lag 1: phi[1][1][*] (head), phi[0][0][*] (tail)
            const float p110 = r1 + x[0] * x[2] + x[1] * x[3];
            const float p111 = i1 + x[0] * x[3] - x[1] * x[2];
            const float p000 = r1 + x[76] * x[78] + x[77] * x[79];
            const float p001 = i1 + x[76] * x[79] - x[77] * x[78];
            // lag 2: phi[0][1][*]
            const float p010 = r2 + x[0] * x[4] + x[1] * x[5];
            const float p011 = i2 + x[0] * x[5] - x[1] * x[4];

            const float dk = p210 * p100 - (p110 * p110 + p111 * p111) / 1.000001f;
            float a1r, a1i, a0r, a0i;
            if (!dk) {
                a1r = 0; a1i = 0;
            } else {
                const float tr = p000 * p110 - p001 * p111 - p010 * p100;
                const float ti = p000 * p111 + p001 * p110 - p011 * p100;
                a1r = tr / dk;
                a1i = ti / dk;
            }
            if (!p100) {
                a0r = 0; a0i = 0;
            } else {
                const float tr = p000 + a1r * p110 + a1i * p111;
                const float ti = p001 + a1i * p110 - a1r * p111;
                a0r = -tr / p100;
                a0i = -ti / p100;
            }
            if (a1r * a1r + a1i * a1i >= 16.0f || a0r * a0r + a0i * a0i >= 16.0f) {
                a1r = 0; a1i = 0; a0r = 0; a0i = 0;
            }
            w.alpha0[lane][0] = a0r; w.alpha0[lane][1] = a0i;
            w.alpha1[lane][0] = a1r; w.alpha1[lane][1] = a1i;
        }
        // ---- sbr_chirp (:1316-1334) ----
        if (lane < n_q) {
            const int m0 = c.bs_invf_mode[0][lane], m1 = c.bs_invf_mode[1][lane];
            float new_bw;
            if (m0 + m1 == 1) new_bw = 0.6f;
            else new_bw = m0 == 0 ? 0.0f : m0 == 1 ? 0.75f : m0 == 2 ? 0.9f : 0.98f;
            const float old = w.bw[lane];
            if (new_bw < old) new_bw = 0.75f    * new_bw + 0.25f    * old;
            else              new_bw = 0.90625f * new_bw + 0.09375f * old;
            w.bw[lane] = new_bw < 0.015625f ? 0.0f : new_bw;
        }
        // envelope index of every time slot
        if (lane < 40) {
            int e = 0;
            for (int q = 1; q < num_env; q++)
                if (lane >= 2 * c.t_env[q]) e = q;
            w.env_of[lane] = e;
        }
        wave_sync();

        // ---- per-band constants of sbr_hf_gen (:1369-1386) ----
        for (int m = lane; m < m_max; m += WAVE) {
            const int k = kx + m;
            int p = -1, base = 0;
            for (int j = 0; j < h.num_patches; j++) {
                const int len = h.patch_num_subbands[j];
                if (m >= base && m < base + len) p = h.patch_start_subband[j] + (m - base);
                base += len;
            }
            int g = -1;
            for (int q = 0; q <= n_q; q++)
                if (k >= h.f_tablenoise[q]) g = q;
            w.kp[m] = p;
            if (p >= 0 && p < 32 && g >= 0) {
                const float b = w.bw[g];
                w.kc[m][0] = w.alpha1[p][0] * b * b;
                w.kc[m][1] = w.alpha1[p][1] * b * b;
                w.kc[m][2] = w.alpha0[p][0] * b;
                w.kc[m][3] = w.alpha0[p][1] * b;
            } else {
                w.kp[m] = -1;
            }
        }

        // ---- sbr_mapping (:1451-1496) ----
        for (int m = lane; m < MAXM; m += WAVE) {
            const int k = kx + m;
            for (int e = 0; e < num_env; e++) {
                uint8_t sidx = 0;
                if (m < m_max) {
                    const int res = c.bs_freq_res[e + 1];
                    const uint8_t *table = res ? h.f_tablehigh : h.f_tablelow;
                    const int ilim = h.n[res];
                    int bi = 0;
                    for (int i = 0; i < ilim; i++)
                        if (k >= table[i]) bi = i;
                    w.e_orig[e][m] = deq_env(w, coupling, ch, e, bi);
                    const int kq = (c.bs_num_noise > 1) && (c.t_env[e] >= c.t_q[1]);
                    int qi = 0;
                    for (int i = 0; i < n_q; i++)
                        if (k >= h.f_tablenoise[i]) qi = i;
                    w.q_map[e][m] = deq_noise(w, coupling, ch, kq, qi);
                    if (c.bs_add_harmonic_flag) {
                        for (int i = 0; i < h.n[1]; i++) {
                            const int mid = (h.f_tablehigh[i] + h.f_tablehigh[i + 1]) >> 1;
                            if (mid == k)
                                sidx = c.bs_add_harmonic[i] *
                                       (e >= c.e_a[1] || (w.s_idx[0][m] == 1));
                        }
                    }
                }
                w.s_idx[e + 1][m] = sidx;
            }
        }
        wave_sync();
        for (int m = lane; m < m_max; m += WAVE) {
            const int k = kx + m;
            for (int e = 0; e < num_env; e++) {
                const int res = c.bs_freq_res[e + 1];
                const uint8_t *table = res ? h.f_tablehigh : h.f_tablelow;
                const int ilim = h.n[res];
                int bi = 0;
                for (int i = 0; i < ilim; i++)
                    if (k >= table[i]) bi = i;
                int present = 0;
                for (int mm = table[bi]; mm < table[bi + 1]; mm++)
                    if (w.s_idx[e + 1][mm - kx]) { present = 1; break; }
                w.s_map[e][m] = (uint8_t)present;
            }
        }

        // ---- sbr_env_estimate (:1499-1546) ----
        if (h.bs_interpol_freq) {
            for (int m = lane; m < m_max; m += WAVE) {
                for (int e = 0; e < num_env; e++) {
                    const float recip_env_size = 0.5f / (c.t_env[e + 1] - c.t_env[e]);
                    const int ilb = c.t_env[e] * 2 + ENV_ADJ, iub = c.t_env[e + 1] * 2 + ENV_ADJ;
                    float sum = 0.0f;
                    for (int i = ilb; i < iub; i++) {
                        float re, im;
                        xhigh(w, m, i, re, im);
                        sum += re * re + im * im;
                    }
                    w.e_curr[e][m] = sum * recip_env_size;
                }
            }
        } else {
            for (int e = 0; e < num_env; e++) {
                const int res = c.bs_freq_res[e + 1];
                const uint8_t *table = res ? h.f_tablehigh : h.f_tablelow;
                const int env_size = 2 * (c.t_env[e + 1] - c.t_env[e]);
                const int ilb = c.t_env[e] * 2 + ENV_ADJ, iub = c.t_env[e + 1] * 2 + ENV_ADJ;
                for (int p = lane; p < h.n[res]; p += WAVE) {
                    float sum = 0.0f;
                    const int den = env_size * (table[p + 1] - table[p]);
                    for (int k = table[p]; k < table[p + 1]; k++)
                        for (int i = ilb; i < iub; i++) {
                            float re, im;
                            xhigh(w, k - kx, i, re, im);
                            sum += re * re + im * im;
                        }
                    sum /= den;
                    for (int k = table[p]; k < table[p + 1]; k++)
                        w.e_curr[e][k - kx] = sum;
                }
            }
        }
        wave_sync();

        // ---- sbr_gain_calc (:1552-1605): one lane per (envelope, limiter band) ----
        // Bands no limiter band covers (last patch dropped, :538-539) keep the
        // zeros of the reference's av_mallocz'ed context.
        for (int t = lane; t < MAXE * MAXM; t += WAVE) {
            (&w.gain[0][0])[t] = 0.0f;
            (&w.q_m[0][0])[t] = 0.0f;
            (&w.s_m[0][0])[t] = 0.0f;
        }
        wave_sync();
        {
            const int n_lim = h.n_lim;
            const float limgain = h.bs_limiter_gains == 0 ? 0.70795f :
                                  h.bs_limiter_gains == 1 ? 1.0f :
                                  h.bs_limiter_gains == 2 ? 1.41254f : 10000000000.0f;
            for (int t = lane; t < num_env * n_lim; t += WAVE) {
                const int e = t / n_lim, kk = t - e * n_lim;
                const int delta = !((e == c.e_a[1]) || (e == c.e_a[0]));
                const int ma = h.f_tablelim[kk] - kx, mb = h.f_tablelim[kk + 1] - kx;
                float sum0 = 0.0f, sum1 = 0.0f;
                for (int m = ma; m < mb; m++) {
                    const float eo = w.e_orig[e][m], qm = w.q_map[e][m], ec = w.e_curr[e][m];
                    const float temp = eo / (1.0f + qm);
                    w.q_m[e][m] = sqrtf(temp * qm);
                    w.s_m[e][m] = sqrtf(temp * (float)w.s_idx[e + 1][m]);
                    if (!w.s_map[e][m])
                        w.gain[e][m] = sqrtf(eo / ((1.0f + ec) * (1.0f + qm * (float)delta)));
                    else
                        w.gain[e][m] = sqrtf(eo * qm / ((1.0f + ec) * (1.0f + qm)));
                }
                for (int m = ma; m < mb; m++) {
                    sum0 += w.e_orig[e][m];
                    sum1 += w.e_curr[e][m];
                }
                float gain_max = limgain * sqrtf((1.1920928955078125e-7f + sum0) / (1.1920928955078125e-7f + sum1));
                gain_max = FFMIN_(100000.0f, gain_max);
                for (int m = ma; m < mb; m++) {
                    const float q_m_max = w.q_m[e][m] * gain_max / w.gain[e][m];
                    w.q_m[e][m]  = FFMIN_(w.q_m[e][m], q_m_max);
                    w.gain[e][m] = FFMIN_(w.gain[e][m], gain_max);
                }
                sum0 = sum1 = 0.0f;
                for (int m = ma; m < mb; m++) {
                    sum0 += w.e_orig[e][m];
                    sum1 += w.e_curr[e][m] * w.gain[e][m] * w.gain[e][m]
                            + w.s_m[e][m] * w.s_m[e][m]
                            + (float)(delta && !w.s_m[e][m]) * w.q_m[e][m] * w.q_m[e][m];
                }
                float gain_boost = sqrtf((1.1920928955078125e-7f + sum0) / (1.1920928955078125e-7f + sum1));
                // FFMIN(1.584893192, gain_boost) is evaluated in double (:1597)
                gain_boost = (float)(1.584893192 > (double)gain_boost ? (double)gain_boost : 1.584893192);
                for (int m = ma; m < mb; m++) {
                    w.gain[e][m] *= gain_boost;
                    w.q_m[e][m]  *= gain_boost;
                    w.s_m[e][m]  *= gain_boost;
                }
            }
        }
        wave_sync();

        // history rows for the smoothing filter (:1630-1639)
        if (reset) {
            for (int t = lane; t < 4 * MAXM; t += WAVE) {
                const int m = t % MAXM;
                if (m < m_max) {
                    (&w.ghist[0][0])[t] = w.gain[0][m];
                    (&w.qhist[0][0])[t] = w.q_m[0][m];
                }
            }
            wave_sync();
        }
    }

    // ---- sbr_hf_assemble (:1608-1714) fused with sbr_x_gen (:1412-1446) ----
    const int t_old = c.t_env_num_env_old;
    const int i_Temp = 2 * t_old - 32 > 0 ? 2 * t_old - 32 : 0;
    const float *ytail_in = st_in + HEAAC_SBR_YTAIL;
    float *ytail_out = st_out + HEAAC_SBR_YTAIL;
    float *X0 = g_X, *X1 = g_X + 38 * 64;
    {
        const int k = lane;                       // one QMF band per lane
        const int m = k - kx;
        const bool hf = start && m >= 0 && m < m_max;
        const float h0 = 0.33333333333333f, h1 = 0.30150283239582f, h2 = 0.21816949906249f,
                    h3 = 0.11516383427084f, h4 = 0.03183050093751f;
        const int phi_sign0 = (1 - 2 * (kx & 1)) * ((m & 1) ? -1 : 1);
        for (int i = 0; i < 38; i++) {
            float yr = 0.0f, yi = 0.0f;
            bool have_y = false;
            if (hf && i >= 2 * t0 && i < 2 * tL) {
                have_y = true;
                const int e = w.env_of[i];
                const bool plain = (e == c.e_a[0]) || (e == c.e_a[1]);
                float xr, xi;
                xhigh(w, m, i + ENV_ADJ, xr, xi);
                float g_filt;
                // g_temp row r holds gain[env_of[r - h_SL]] for r >= h_SL + 2 t0,
                // the 4 history rows below that
#define GROW(arr, hist, r) ((r) >= h_SL + 2 * t0 ? w.arr[w.env_of[(r) - h_SL]][m] : w.hist[(r) - 2 * t0][m])
                if (h_SL && !plain) {
                    const int idx1 = i + h_SL;
                    g_filt = 0.0f;
                    g_filt += GROW(gain, ghist, idx1 - 0) * h0;
                    g_filt += GROW(gain, ghist, idx1 - 1) * h1;
                    g_filt += GROW(gain, ghist, idx1 - 2) * h2;
                    g_filt += GROW(gain, ghist, idx1 - 3) * h3;
                    g_filt += GROW(gain, ghist, idx1 - 4) * h4;
                } else {
                    g_filt = GROW(gain, ghist, i + h_SL);
                }
                yr = xr * g_filt;
                yi = xi * g_filt;
                const int slot = i - 2 * t0;
                const int isine = (idxsine + slot) & 3;
                const int phi_re = isine == 0 ? 1 : isine == 2 ? -1 : 0;
                const int phi_im = isine == 1 ? 1 : isine == 3 ? -1 : 0;
                const float sm = w.s_m[e][m];
                if (!plain) {
                    if (sm) {
                        yr += sm * (float)phi_re;
                        yi += sm * (float)(phi_im * phi_sign0);
                    } else {
                        float q_filt;
                        if (h_SL) {
                            const int idx1 = i + h_SL;
                            q_filt = 0.0f;
                            q_filt += GROW(q_m, qhist, idx1 - 0) * h0;
                            q_filt += GROW(q_m, qhist, idx1 - 1) * h1;
                            q_filt += GROW(q_m, qhist, idx1 - 2) * h2;
                            q_filt += GROW(q_m, qhist, idx1 - 3) * h3;
                            q_filt += GROW(q_m, qhist, idx1 - 4) * h4;
                        } else {
                            q_filt = w.q_m[e][m];          // q_temp[i][m], h_SL == 0
                        }
                        const unsigned in = (idxnoise + (unsigned)slot * m_max + m + 1) & 0x1ff;
                        yr += q_filt * g_noise[2 * in];
                        yi += q_filt * g_noise[2 * in + 1];
                    }
                } else {
                    yr += sm * (float)phi_re;
                    yi += sm * (float)(phi_im * phi_sign0);
                }
#undef GROW
            }
            // ytail: Y[1][32..37]
            if (i >= 32) {
                const int o = ((i - 32) * 64 + k) * 2;
                if (have_y) { ytail_out[o] = yr; ytail_out[o + 1] = yi; }
                else if (ytail_out != ytail_in) { ytail_out[o] = ytail_in[o]; ytail_out[o + 1] = ytail_in[o + 1]; }
            }
            // x_gen
            float xr = 0.0f, xi = 0.0f;
            if (i < i_Temp) {
                if (k < kx_old) {
                    if (k < 32) { xr = w.xlow[k * XL_STRIDE + 2 * (i + ENV_ADJ)]; xi = w.xlow[k * XL_STRIDE + 2 * (i + ENV_ADJ) + 1]; }
                } else if (k < kx_old + m_old) {
                    xr = ytail_in[(i * 64 + k) * 2]; xi = ytail_in[(i * 64 + k) * 2 + 1];
                }
            } else {
                if (k < kx) {
                    if (k < 32) { xr = w.xlow[k * XL_STRIDE + 2 * (i + ENV_ADJ)]; xi = w.xlow[k * XL_STRIDE + 2 * (i + ENV_ADJ) + 1]; }
                } else if (k < kx + m_max && i < 32) {
                    xr = yr; xi = yi;
                }
            }
            X0[i * 64 + k] = xr;
            X1[i * 64 + k] = xi;
        }
    }

    // ---- remaining state ----
    if (start) {
        if (lane < 5) st_out[HEAAC_SBR_BW + lane] = w.bw[lane];
        if (lane == 0) {
            const unsigned slots = 2 * (tL - t0);
            st_out[HEAAC_SBR_IDXNOISE] = __uint_as_float((idxnoise + slots * m_max) & 0x1ff);
            st_out[HEAAC_SBR_IDXSINE]  = __uint_as_float((idxsine + slots) & 3);
        }
        if (lane < 12) {
            // s_indexmapped[0] <- s_indexmapped[bs_num_env]
            reinterpret_cast<uint32_t *>(st_out + HEAAC_SBR_SIDX)[lane] =
                reinterpret_cast<const uint32_t *>(&w.s_idx[num_env][0])[lane];
        }
        if (h_SL) {
            for (int t = lane; t < 4 * MAXM; t += WAVE) {
                const int j = t / MAXM, m = t % MAXM;
                float g = 0.0f, q = 0.0f;
                if (m < m_max) {
                    const int r = 2 * tL + j;           // g_temp row, >= h_SL + 2 t0
                    const int e = w.env_of[r - h_SL];
                    g = w.gain[e][m];
                    q = w.q_m[e][m];
                }
                st_out[HEAAC_SBR_GTAIL + t] = g;
                st_out[HEAAC_SBR_QTAIL + t] = q;
            }
        } else if (st_out != st_in) {
            for (int t = lane; t < 4 * MAXM; t += WAVE) {
                st_out[HEAAC_SBR_GTAIL + t] = st_in[HEAAC_SBR_GTAIL + t];
                st_out[HEAAC_SBR_QTAIL + t] = st_in[HEAAC_SBR_QTAIL + t];
            }
        }
    } else if (st_out != st_in) {
        for (int t = HEAAC_SBR_GTAIL + lane; t < HEAAC_ST_SBR; t += WAVE)
            st_out[t] = st_in[t];
    }
    if (lane == 0 && st_out != st_in) st_out[HEAAC_SBR_PAD] = st_in[HEAAC_SBR_PAD];
    wave_sync();
}

__global__ __launch_bounds__(HF_WAVES * WAVE)
void k_hfadj(const float *__restrict__ g_tab,
             const HeaacSbrFrame *__restrict__ g_sbr, const HeaacSbrHeader *__restrict__ g_hdr,
             const float *g_W, const float *g_state_in, float *g_state_out, int state_words,
             int ncore, int off_sbr0, float *g_X, unsigned long long n_units)
{
    __shared__ HfWave S[HF_WAVES];
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
    for (unsigned long long u = (unsigned long long)blockIdx.x * HF_WAVES + wave; u < n_units;
         u += (unsigned long long)gridDim.x * HF_WAVES) {
        const unsigned long long f = u / ncore;
        const int ch = (int)(u - f * ncore);
        const int off = off_sbr0 + ch * HEAAC_ST_SBR;
        hf_channel(S[wave], g_tab + TB_NOISE, &g_sbr[f], g_hdr, ch, g_W + u * 2048,
                   g_state_in + f * state_words + off, g_state_out + f * state_words + off,
                   g_X + (f * 2 + ch) * (2 * 38 * 64), lane);
    }
}

// ===========================================================================
// K_D  QMF synthesis (64 bands, div = 0) + output, one wave per frame
// ===========================================================================
#define SYN_WAVES 6
#define VB_STRIDE 129             // v slot row: 128 + 1 pad
#define VB_ROWS   41              // 32 new slots (newest first) + 9 history slots

struct SynWave {
    float vb[VB_ROWS * VB_STRIDE];
    uint16_t pcm0[2048];
};
struct SynLds {
    float win[640];               // sbr_qmf_window_us
    float rot[64];                // SBR synthesis MDCT (scale 1/64): tcos[32], tsin[32]
    float c16[8], c32[12];
    SynWave w[SYN_WAVES];
};

// swap with the neighbouring lane (lane ^ 1): DPP quad_perm [1,0,3,2]
__device__ __forceinline__ float lane_xor1(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
}

// sbr_qmf_synthesis (aacsbr.c:1175-1230), div = 0, for one channel.
//   X0/X1 : re / im planes [38][64] (slots 0..31 used)
//   v_in/v_out : 1152-float ring state, newest slot first
//   emit(i, n, value) receives out[64 i + n]
template <class Emit>
__device__ __forceinline__ void synth_channel(const SynLds &S, SynWave &w, const float *X0, const float *X1,
                                              const float *v_in, float *v_out,
                                              float scale, float bias, int lane, Emit emit)
{
    // history: 9 slots behind the 32 new ones
    for (int t = lane; t < 1152; t += WAVE)
        w.vb[(32 + (t >> 7)) * VB_STRIDE + (t & 127)] = v_in[t];
    // 64 IMDCTs (N = 128): lane = (slot, re/im plane)
    {
        const int i = lane >> 1, part = lane & 1;
        const float *row = (part ? X1 : X0) + i * 64;
        float o[64];
        if (part) {
            // X[1][i][n] = -X[1][i][n] for odd n (:1201-1203)
            imdct128_reg([&](int j) -> float { return (j & 1) ? -row[j] : row[j]; }, o, S.rot, S.c16, S.c32);
        } else {
            imdct128_reg([&](int j) -> float { return row[j]; }, o, S.rot, S.c16, S.c32);
        }
        // v[n] = -buf0[63-n] + buf1[n];  v[127-n] = buf0[63-n] + buf1[n]   (:1206-1209)
        // even lane holds buf0 and produces v[0..63], odd lane holds buf1 and
        // produces v[64..127].
        float *vs = w.vb + (31 - i) * VB_STRIDE;
#pragma unroll
        for (int n = 0; n < 64; n++) {
            // both lanes exchange the element the partner needs for index n
            const float mine = part ? o[n] : o[63 - n];        // buf1[n] | buf0[63-n]
            const float other = lane_xor1(mine);               // buf0[63-n] | buf1[n]
            if (part) vs[127 - n] = other + mine;              //  buf0[63-n] + buf1[n]
            else      vs[n] = -mine + other;                   // -buf0[63-n] + buf1[n]
        }
    }
    wave_sync();
    // 10-tap polyphase sum (:1210-1219), lane = n
    {
        const int n = lane;
        float wt[10];
#pragma unroll
        for (int j = 0; j < 10; j++) wt[j] = S.win[64 * j + n];
        const bool scale_and_bias = scale != 1.0f || bias != 0.0f;
        for (int i = 0; i < 32; i++) {
            const float *v = w.vb + (31 - i) * VB_STRIDE + n;
            float acc = v[0] * wt[0] + 0.0f;
            acc = v[1 * VB_STRIDE + 64] * wt[1] + acc;
            acc = v[2 * VB_STRIDE]      * wt[2] + acc;
            acc = v[3 * VB_STRIDE + 64] * wt[3] + acc;
            acc = v[4 * VB_STRIDE]      * wt[4] + acc;
            acc = v[5 * VB_STRIDE + 64] * wt[5] + acc;
            acc = v[6 * VB_STRIDE]      * wt[6] + acc;
            acc = v[7 * VB_STRIDE + 64] * wt[7] + acc;
            acc = v[8 * VB_STRIDE]      * wt[8] + acc;
            acc = v[9 * VB_STRIDE + 64] * wt[9] + acc;
            if (scale_and_bias) acc = acc * scale + bias;
            emit(i, n, acc);
        }
    }
    // new ring state: slots 31..23
    for (int t = lane; t < 1152; t += WAVE)
        v_out[t] = w.vb[(t >> 7) * VB_STRIDE + (t & 127)];
    wave_sync();
}

template <int FMT>
__global__ __launch_bounds__(SYN_WAVES * WAVE)
void k_synth(const float *__restrict__ g_tab, const float *g_X,
             const float *g_state_in, float *g_state_out, int state_words, int off_syn0,
             int nout, int copy_mono, void *__restrict__ g_pcm, float scale, float bias,
             unsigned long long n_frames, unsigned long long pcm_frame0)
{
    __shared__ SynLds S;
    for (int i = threadIdx.x; i < 640; i += blockDim.x) S.win[i] = g_tab[TB_QMF_US + i];
    if (threadIdx.x < 64) S.rot[threadIdx.x] = g_tab[TB_ROT128S + threadIdx.x];
    if (threadIdx.x < 5) S.c16[threadIdx.x] = g_tab[TB_COS16 + threadIdx.x];
    if (threadIdx.x < 9) S.c32[threadIdx.x] = g_tab[TB_COS32 + threadIdx.x];
    __syncthreads();
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
    SynWave &w = S.w[wave];
    for (unsigned long long f = (unsigned long long)blockIdx.x * SYN_WAVES + wave; f < n_frames;
         f += (unsigned long long)gridDim.x * SYN_WAVES) {
        const float *st_in = g_state_in + f * state_words + off_syn0;
        float *st_out = g_state_out + f * state_words + off_syn0;
        for (int ch = 0; ch < nout; ch++) {
            // copy_mono: ps->start == 0 handled by the caller pointing both
            // channels at plane set 0 (aacsbr.c:1755)
            const float *X0 = g_X + (f * 2 + (copy_mono ? 0 : ch)) * (2 * 38 * 64);
            const float *X1 = X0 + 38 * 64;
            const float *v_in = st_in + ch * HEAAC_ST_SYNTH;
            float *v_out = st_out + ch * HEAAC_ST_SYNTH;
            if (FMT == HEAAC_PCM_F32_PLANAR) {
                float *o = reinterpret_cast<float *>(g_pcm) + ((pcm_frame0 + f) * nout + ch) * 2048;
                synth_channel(S, w, X0, X1, v_in, v_out, scale, bias, lane,
                              [&](int i, int n, float v) { o[64 * i + n] = v; });
            } else if (nout == 1) {
                int16_t *o = reinterpret_cast<int16_t *>(g_pcm) + (pcm_frame0 + f) * 2048;
                synth_channel(S, w, X0, X1, v_in, v_out, scale, bias, lane,
                              [&](int i, int n, float v) { o[64 * i + n] = (int16_t)float_to_int16_one(v); });
            } else if (ch == 0) {
                synth_channel(S, w, X0, X1, v_in, v_out, scale, bias, lane,
                              [&](int i, int n, float v) { w.pcm0[64 * i + n] = (uint16_t)float_to_int16_one(v); });
            } else {
                uint32_t *o = reinterpret_cast<uint32_t *>(g_pcm) + (pcm_frame0 + f) * 2048;
                synth_channel(S, w, X0, X1, v_in, v_out, scale, bias, lane,
                              [&](int i, int n, float v) {
                                  o[64 * i + n] = (uint32_t)w.pcm0[64 * i + n] |
                                                  ((uint32_t)(float_to_int16_one(v) & 0xffff) << 16);
                              });
            }
        }
    }
}

// Stage-level batched filterbanks (heaac_qmf_analysis_batch / _synthesis_batch)
__global__ __launch_bounds__(ANA_WAVES * WAVE)
void k_qmf_analysis(const float *__restrict__ g_tab, const float *__restrict__ g_in,
                    const float *g_xh_in, float *g_xh_out, float *__restrict__ g_W, float scale,
                    unsigned long long n)
{
    __shared__ float qmf_ds[320];
    __shared__ float rot[64];
    __shared__ float c16[8], c32[12];
    __shared__ float pool[ANA_WAVES][ANA_POOL];
    for (int i = threadIdx.x; i < 320; i += blockDim.x) qmf_ds[i] = g_tab[TB_QMF_DS + i];
    if (threadIdx.x < 64) rot[threadIdx.x] = g_tab[TB_ROT128A + threadIdx.x];
    if (threadIdx.x < 5) c16[threadIdx.x] = g_tab[TB_COS16 + threadIdx.x];
    if (threadIdx.x < 9) c32[threadIdx.x] = g_tab[TB_COS32 + threadIdx.x];
    __syncthreads();
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
    float *p = pool[wave];
    for (unsigned long long u = (unsigned long long)blockIdx.x * ANA_WAVES + wave; u < n;
         u += (unsigned long long)gridDim.x * ANA_WAVES) {
        float *x = p + 2080;
        for (int i = lane; i < 288; i += WAVE) x[i] = g_xh_in[u * 288 + i];
        if (scale != 1.0f) {
            for (int i = lane; i < 1024; i += WAVE) x[288 + i] = g_in[u * 1024 + i] * scale;
        } else {
            for (int i = lane; i < 1024; i += WAVE) x[288 + i] = g_in[u * 1024 + i];
        }
        wave_sync();
        for (int i = lane; i < 288; i += WAVE) g_xh_out[u * 288 + i] = x[1024 + i];
        qmf_analysis_wave(qmf_ds, rot, c16, c32, x, p, g_W + u * 2048, lane);
    }
}

__global__ __launch_bounds__(SYN_WAVES * WAVE)
void k_qmf_synthesis(const float *__restrict__ g_tab, const float *__restrict__ g_X /* [n][2][32][64] */,
                     const float *g_v_in, float *g_v_out, float *__restrict__ g_out,
                     float scale, float bias, unsigned long long n)
{
    __shared__ SynLds S;
    for (int i = threadIdx.x; i < 640; i += blockDim.x) S.win[i] = g_tab[TB_QMF_US + i];
    if (threadIdx.x < 64) S.rot[threadIdx.x] = g_tab[TB_ROT128S + threadIdx.x];
    if (threadIdx.x < 5) S.c16[threadIdx.x] = g_tab[TB_COS16 + threadIdx.x];
    if (threadIdx.x < 9) S.c32[threadIdx.x] = g_tab[TB_COS32 + threadIdx.x];
    __syncthreads();
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
    for (unsigned long long u = (unsigned long long)blockIdx.x * SYN_WAVES + wave; u < n;
         u += (unsigned long long)gridDim.x * SYN_WAVES) {
        const float *X0 = g_X + u * 4096, *X1 = X0 + 2048;
        float *o = g_out + u * 2048;
        synth_channel(S, S.w[wave], X0, X1, g_v_in + u * 1152, g_v_out + u * 1152, scale, bias, lane,
                      [&](int i, int nn, float v) { o[64 * i + nn] = v; });
    }
}

// ===========================================================================
// host side: launch the HE pipeline over one chunk of frames
// ===========================================================================
static int he_grid(unsigned long long units, int per_block)
{
    unsigned long long g = (units + per_block - 1) / per_block;
    if (g > 256) g = 256;           // one persistent workgroup per CU (LDS-bound kernels)
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" int heaac_launch_he(const float *d_tab, const uint16_t *d_rev, int cfg,
                               const float *d_coeffs, const HeaacIcs *d_ics,
                               const HeaacSbrFrame *d_sbr, const HeaacSbrHeader *d_hdr,
                               const HeaacPsFrame *d_ps,
                               const float *d_state_in, float *d_state_out,
                               void *d_pcm, int pcm_format,
                               float *d_ws_W, float *d_ws_X,
                               size_t n, size_t pcm_frame0, hipStream_t s)
{
    const int ncore = cfg == HEAAC_CFG_HEV1 ? 2 : 1;
    const int nout  = cfg == HEAAC_CFG_HEV1_MONO ? 1 : 2;
    int words, off_saved0 = 0, off_sbr0, off_syn0;
    if (cfg == HEAAC_CFG_HEV1) {
        words = HEAAC_STATE_WORDS_HEV1; off_sbr0 = 2 * HEAAC_ST_SAVED; off_syn0 = off_sbr0 + 2 * HEAAC_ST_SBR;
    } else if (cfg == HEAAC_CFG_HEV1_MONO) {
        words = HEAAC_STATE_WORDS_HEV1_MONO; off_sbr0 = HEAAC_ST_SAVED; off_syn0 = off_sbr0 + HEAAC_ST_SBR;
    } else if (cfg == HEAAC_CFG_HEV2) {
        words = HEAAC_STATE_WORDS_HEV2; off_sbr0 = HEAAC_ST_SAVED; off_syn0 = off_sbr0 + HEAAC_ST_SBR;
    } else
        return HEAAC_ERR_ARG;
    const unsigned long long units = (unsigned long long)n * ncore;
    const float sf_scale = HEAAC_SF_SCALE;

    hipLaunchKernelGGL(k_core_ana, dim3(he_grid(units, ANA_WAVES)), dim3(ANA_WAVES * WAVE), 0, s,
                       d_tab, d_rev, d_coeffs, d_ics, d_state_in, d_state_out, words, ncore,
                       off_saved0, off_sbr0, d_ws_W, 1 / (-1024 * sf_scale), units);
    hipLaunchKernelGGL(k_hfadj, dim3(he_grid(units, HF_WAVES)), dim3(HF_WAVES * WAVE), 0, s,
                       d_tab, d_sbr, d_hdr, d_ws_W, d_state_in, d_state_out, words, ncore, off_sbr0,
                       d_ws_X, units);
    int copy_mono = 0;
    if (cfg == HEAAC_CFG_HEV2) {
        int rc = heaac_launch_ps(d_tab, d_ps, d_sbr, d_hdr, d_state_in, d_state_out, words,
                                 off_syn0 + 2 * HEAAC_ST_SYNTH, d_ws_X, n, s);
        if (rc != HEAAC_OK) return rc;
    }
    const float scale = -1024 * sf_scale, bias = HEAAC_ADD_BIAS;
    const dim3 g(he_grid(n, SYN_WAVES)), b(SYN_WAVES * WAVE);
    if (pcm_format == HEAAC_PCM_F32_PLANAR)
        hipLaunchKernelGGL((k_synth<HEAAC_PCM_F32_PLANAR>), g, b, 0, s, d_tab, d_ws_X, d_state_in, d_state_out,
                           words, off_syn0, nout, copy_mono, d_pcm, scale, bias,
                           (unsigned long long)n, (unsigned long long)pcm_frame0);
    else if (pcm_format == HEAAC_PCM_S16_INTERLEAVED)
        hipLaunchKernelGGL((k_synth<HEAAC_PCM_S16_INTERLEAVED>), g, b, 0, s, d_tab, d_ws_X, d_state_in, d_state_out,
                           words, off_syn0, nout, copy_mono, d_pcm, scale, bias,
                           (unsigned long long)n, (unsigned long long)pcm_frame0);
    else
        return HEAAC_ERR_ARG;
    return hipGetLastError() == hipSuccess ? HEAAC_OK : HEAAC_ERR_HIP;
}

extern "C" int heaac_launch_qmf_analysis(const float *d_tab, const float *d_in, const float *d_xh_in,
                                         float *d_xh_out, float *d_W, float scale, size_t n, hipStream_t s)
{
    if (!n) return HEAAC_OK;
    hipLaunchKernelGGL(k_qmf_analysis, dim3(he_grid(n, ANA_WAVES)), dim3(ANA_WAVES * WAVE), 0, s,
                       d_tab, d_in, d_xh_in, d_xh_out, d_W, scale, (unsigned long long)n);
    return hipGetLastError() == hipSuccess ? HEAAC_OK : HEAAC_ERR_HIP;
}

extern "C" int heaac_launch_qmf_synthesis(const float *d_tab, const float *d_X, const float *d_v_in,
                                          float *d_v_out, float *d_out, float scale, float bias,
                                          size_t n, hipStream_t s)
{
    if (!n) return HEAAC_OK;
    hipLaunchKernelGGL(k_qmf_synthesis, dim3(he_grid(n, SYN_WAVES)), dim3(SYN_WAVES * WAVE), 0, s,
                       d_tab, d_X, d_v_in, d_v_out, d_out, scale, bias, (unsigned long long)n);
    return hipGetLastError() == hipSuccess ? HEAAC_OK : HEAAC_ERR_HIP;
}
