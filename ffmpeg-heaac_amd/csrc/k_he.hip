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

#ifdef HF_STAMPS
__device__ unsigned long long g_hf_stamps[16];
#define HSTAMP(i) do { wave_sync(); if (lane == 0 && blockIdx.x == 7 && threadIdx.x < 64) g_hf_stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define HSTAMP(i) do {} while (0)
#endif

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
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE), lane = threadIdx.x % WAVE;
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
// Lane = QMF band k (m = k - kx for the SBR range).  Everything that is per band
// and per envelope (mapped scalefactors, estimated envelope, gains) lives in that
// lane's registers; the only cross-lane steps are the limiter-band sums of
// sbr_gain_calc, which go through small LDS arrays in the reference's order.
#define HF_WAVES 8
#define XL_STRIDE 81              // X_low row: 40 slots * (re,im) + 1 pad (bank spread)
#define MAXM 48                   // e_origmapped[7][48] etc. in the reference (sbr.h:165-177)
#define MAXE 5

struct HfWave {
    float xlow[32 * XL_STRIDE];   // X_low[k][i][re,im]
    float alpha0[32][2], alpha1[32][2];
    float bw[8];
    float sumA[MAXE][MAXM], sumB[MAXE][MAXM];     // per-band terms of the limiter-band sums
    float bandv[MAXE][32];                        // gain_max / gain_boost per (envelope, limiter band)
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

#define FFMIN_(a, b) ((a) > (b) ? (b) : (a))

// X_high[k][idx] from three consecutive X_low samples of the patch source band
// (sbr_hf_gen, aacsbr.c:1388-1402); x2 = X_low[p][idx-2], x1 = [idx-1], x0 = [idx].
__device__ __forceinline__ void xhigh3(float2 x2, float2 x1, float2 x0, const float *a, float &re, float &im)
{
    re = x2.x * a[0] - x2.y * a[1] + x1.x * a[2] - x1.y * a[3] + x0.x;
    im = x2.y * a[0] + x2.x * a[1] + x1.y * a[2] + x1.x * a[3] + x0.y;
}

__device__ __forceinline__ void hf_channel(HfWave &w, const float *g_noise /* LDS */,
                                           const HeaacSbrFrame *g_fr, const HeaacSbrHeader *g_hdr,
                                           int ch, const float *g_W,
                                           const float *st_in, float *st_out,
                                           float *g_X /* [2][38][64] */, int lane)
{
    HSTAMP(0);
    // ---- issue every global load up front: parameters, W, state ----
    const int hdr_idx = g_fr->hdr;
    // channel records (2 x 336 B = 168 dwords) and the header (532 B = 133 dwords):
    // all loads issued before any LDS store
    uint32_t creg[3], hreg[3];
    {
        const uint32_t *cs = reinterpret_cast<const uint32_t *>(&g_fr->ch[0]);
        const uint32_t *hs_ = reinterpret_cast<const uint32_t *>(&g_hdr[hdr_idx]);
#pragma unroll
        for (int r = 0; r < 3; r++) {
            creg[r] = lane + 64 * r < 168 ? cs[lane + 64 * r] : 0;
            hreg[r] = lane + 64 * r < 133 ? hs_[lane + 64 * r] : 0;
        }
    }
    const int start = g_fr->start, reset = g_fr->reset;
    const int kx_old = g_fr->kx_old, m_old = g_fr->m_old;
    const int coupling = g_fr->bs_coupling;
    float2 wreg[16], treg[4];
    {
        const float2 *W2 = reinterpret_cast<const float2 *>(g_W);
        const float2 *T2 = reinterpret_cast<const float2 *>(st_in + HEAAC_SBR_WTAIL);
#pragma unroll
        for (int r = 0; r < 16; r++) wreg[r] = W2[lane + 64 * r];
#pragma unroll
        for (int r = 0; r < 4; r++) treg[r] = T2[lane + 64 * r];
    }
    const int k = lane;                                  // this lane's QMF band
    float ghist[4], qhist[4];                            // g_temp / q_temp history rows of band m
    unsigned idxnoise = __float_as_uint(st_in[HEAAC_SBR_IDXNOISE]);
    unsigned idxsine  = __float_as_uint(st_in[HEAAC_SBR_IDXSINE]);
    const float bw_in = lane < 5 ? st_in[HEAAC_SBR_BW + lane] : 0.0f;
    {
        uint32_t *cd = reinterpret_cast<uint32_t *>(&w.c[0]);
        uint32_t *hd = reinterpret_cast<uint32_t *>(&w.h);
#pragma unroll
        for (int r = 0; r < 3; r++) {
            if (lane + 64 * r < 168) cd[lane + 64 * r] = creg[r];
            if (lane + 64 * r < 133) hd[lane + 64 * r] = hreg[r];
        }
    }
    wave_sync();
    const HeaacSbrHeader &h = w.h;
    const HeaacSbrChannel &c = w.c[ch];
    const int kx = h.kx, m_max = h.m, n_q = h.n_q;
    const int m = k - kx;
    const bool in_sbr = m >= 0 && m < m_max && m < MAXM;
    const int num_env = c.bs_num_env;
    const int t0 = c.t_env[0], tL = c.t_env[num_env];
    const int h_SL = 4 * !h.bs_smoothing_mode;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        ghist[j] = in_sbr ? st_in[HEAAC_SBR_GTAIL + j * MAXM + m] : 0.0f;
        qhist[j] = in_sbr ? st_in[HEAAC_SBR_QTAIL + j * MAXM + m] : 0.0f;
    }
    const int sidx0 = in_sbr ? reinterpret_cast<const uint8_t *>(st_in + HEAAC_SBR_SIDX)[m] : 0;
    if (reset) idxnoise = 0;                     // sbr_make_f_derived, :587-588

    HSTAMP(1);
    // ---- sbr_lf_gen (:1337-1357): W -> X_low, previous tail for slots 0..7 ----
    {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int t = lane + 64 * r, i = t >> 5, kk = t & 31;
            float2 v = wreg[r];
            if (kk >= kx) v = make_float2(0.0f, 0.0f);
            float *d = w.xlow + kk * XL_STRIDE + 2 * (i + 8);
            d[0] = v.x; d[1] = v.y;
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int t = lane + 64 * r, i = t >> 5, kk = t & 31;
            float2 v = treg[r];
            if (kk >= kx_old) v = make_float2(0.0f, 0.0f);
            float *d = w.xlow + kk * XL_STRIDE + 2 * i;
            d[0] = v.x; d[1] = v.y;
        }
        // new tail = W[1][24..31]: registers 12..15 hold slots 24..31
        float2 *To = reinterpret_cast<float2 *>(st_out + HEAAC_SBR_WTAIL);
#pragma unroll
        for (int r = 0; r < 4; r++) To[lane + 64 * r] = wreg[12 + r];
    }
    if (lane < 8) w.bw[lane] = bw_in;
    wave_sync();

    // per-lane registers of the envelope adjuster
    float e_orig[MAXE], q_map[MAXE], e_curr[MAXE], gain[MAXE], q_m[MAXE], s_m[MAXE];
    int sidx[MAXE];                               // s_indexmapped[e + 1][m]
    unsigned smap = 0;                            // bit e: s_mapped[e][m]
    float kc[4] = { 0, 0, 0, 0 };                 // hf_gen alpha[0..3]
#pragma unroll
    for (int e = 0; e < MAXE; e++) { e_orig[e] = q_map[e] = e_curr[e] = gain[e] = q_m[e] = s_m[e] = 0.0f; sidx[e] = 0; }
    // X_low row this lane reads: its own band below kx, the patch source above
    const int p_src = in_sbr ? (int)h.map_src[k] : 0xff;
    const bool has_src = p_src < 32;
    const int row = k < kx ? (k < 32 ? k : 0) : (has_src ? p_src : 0);
    const float2 *xr = reinterpret_cast<const float2 *>(0);   // (unaligned rows: read as two floats)
    (void)xr;
    const float *xrow = w.xlow + row * XL_STRIDE;

    if (start) {
        HSTAMP(2);
        // ---- sbr_hf_inverse_filter (:1261-1313) + autocorrelate (:1232-1255) ----
        if (lane < h.k0 && lane < 32) {
            // the whole row first (all LDS reads in flight), then the five running sums
            float x[80];
            const float *xs = w.xlow + lane * XL_STRIDE;
#pragma unroll
            for (int i = 0; i < 80; i++) x[i] = xs[i];
            float r0 = 0.0f, r1 = 0.0f, i1 = 0.0f, r2 = 0.0f, i2 = 0.0f;
#pragma unroll
            for (int i = 1; i < 38; i++) {
                const float a = x[2 * i], b = x[2 * i + 1];
                r0 += a * a + b * b;
                r1 += a * x[2 * i + 2] + b * x[2 * i + 3];
                i1 += a * x[2 * i + 3] - b * x[2 * i + 2];
                r2 += a * x[2 * i + 4] + b * x[2 * i + 5];
                i2 += a * x[2 * i + 5] - b * x[2 * i + 4];
            }
            const float p210 = r0 + x[0] * x[0] + x[1] * x[1];
            const float p100 = r0 + x[76] * x[76] + x[77] * x[77];
            const float p110 = r1 + x[0] * x[2] + x[1] * x[3];
            const float p111 = i1 + x[0] * x[3] - x[1] * x[2];
            const float p000 = r1 + x[76] * x[78] + x[77] * x[79];
            const float p001 = i1 + x[76] * x[79] - x[77] * x[78];
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
        HSTAMP(3);
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
        wave_sync();

        HSTAMP(4);
        // ---- per-band constants of sbr_hf_gen (:1369-1386) ----
        if (has_src) {
            const int g = h.map_nq[k];
            const float b = w.bw[g < 5 ? g : 0];
            kc[0] = w.alpha1[p_src][0] * b * b;
            kc[1] = w.alpha1[p_src][1] * b * b;
            kc[2] = w.alpha0[p_src][0] * b;
            kc[3] = w.alpha0[p_src][1] * b;
        }

        HSTAMP(5);
        // ---- sbr_mapping (:1451-1496) ----
        if (in_sbr) {
            const int hi = h.map_hi[k], lo = h.map_lo[k], nq = h.map_nq[k], mid = h.map_mid[k];
#pragma unroll
            for (int e = 0; e < MAXE; e++) {
                if (e < num_env) {
                    const int res = c.bs_freq_res[e + 1];
                    e_orig[e] = deq_env(w, coupling, ch, e, res ? hi : lo);
                    const int kq = (c.bs_num_noise > 1) && (c.t_env[e] >= c.t_q[1]);
                    q_map[e] = deq_noise(w, coupling, ch, kq, nq);
                    if (c.bs_add_harmonic_flag && mid != 0xff)
                        sidx[e] = c.bs_add_harmonic[mid] * (e >= c.e_a[1] || (sidx0 == 1));
                }
            }
        }
        // s_mapped[e][m]: any sinusoid inside the band of the envelope's resolution (:1479-1491)
#pragma unroll
        for (int e = 0; e < MAXE; e++) {
            if (e < num_env) {
                const unsigned long long present = __ballot(sidx[e] != 0);      // bit = lane = band k
                if (in_sbr) {
                    const int res = c.bs_freq_res[e + 1];
                    const uint8_t *table = res ? h.f_tablehigh : h.f_tablelow;
                    const int bi = res ? h.map_hi[k] : h.map_lo[k];
                    const int lo_k = table[bi], hi_k = table[bi + 1];               // [lo_k, hi_k)
                    const unsigned long long mask = (hi_k >= 64 ? ~0ull : ((1ull << hi_k) - 1)) & ~((1ull << lo_k) - 1);
                    if (present & mask) smap |= 1u << e;
                }
            }
        }

        HSTAMP(6);
        // ---- sbr_env_estimate (:1499-1546) ----
        if (h.bs_interpol_freq) {
            if (in_sbr) {
#pragma unroll
                for (int e = 0; e < MAXE; e++) {
                    if (e < num_env) {
                        const float recip_env_size = 0.5f / (c.t_env[e + 1] - c.t_env[e]);
                        const int ilb = c.t_env[e] * 2 + ENV_ADJ, iub = c.t_env[e + 1] * 2 + ENV_ADJ;
                        float sum = 0.0f;
                        if (has_src) {
                            float2 x2 = make_float2(xrow[2 * (ilb - 2)], xrow[2 * (ilb - 2) + 1]);
                            float2 x1 = make_float2(xrow[2 * (ilb - 1)], xrow[2 * (ilb - 1) + 1]);
                            for (int i = ilb; i < iub; i++) {
                                const float2 x0 = make_float2(xrow[2 * i], xrow[2 * i + 1]);
                                float re, im;
                                xhigh3(x2, x1, x0, kc, re, im);
                                sum += re * re + im * im;
                                x2 = x1; x1 = x0;
                            }
                        } else {
                            for (int i = ilb; i < iub; i++) sum += 0.0f * 0.0f + 0.0f * 0.0f;
                        }
                        e_curr[e] = sum * recip_env_size;
                    }
                }
            }
        } else {
            // one lane per band of the envelope's frequency table; result broadcast through LDS
#pragma unroll
            for (int e = 0; e < MAXE; e++) {
                if (e < num_env) {
                    const int res = c.bs_freq_res[e + 1];
                    const uint8_t *table = res ? h.f_tablehigh : h.f_tablelow;
                    const int env_size = 2 * (c.t_env[e + 1] - c.t_env[e]);
                    const int ilb = c.t_env[e] * 2 + ENV_ADJ, iub = c.t_env[e + 1] * 2 + ENV_ADJ;
                    if (lane < h.n[res]) {
                        float sum = 0.0f;
                        const int den = env_size * (table[lane + 1] - table[lane]);
                        for (int kk = table[lane]; kk < table[lane + 1]; kk++) {
                            const int ps = h.map_src[kk];
                            float a[4] = { 0, 0, 0, 0 };
                            if (ps < 32) {
                                const int g = h.map_nq[kk];
                                const float b = w.bw[g < 5 ? g : 0];
                                a[0] = w.alpha1[ps][0] * b * b; a[1] = w.alpha1[ps][1] * b * b;
                                a[2] = w.alpha0[ps][0] * b;     a[3] = w.alpha0[ps][1] * b;
                            }
                            const float *xs = w.xlow + (ps < 32 ? ps : 0) * XL_STRIDE;
                            for (int i = ilb; i < iub; i++) {
                                float re = 0.0f, im = 0.0f;
                                if (ps < 32)
                                    xhigh3(make_float2(xs[2 * i - 4], xs[2 * i - 3]), make_float2(xs[2 * i - 2], xs[2 * i - 1]),
                                           make_float2(xs[2 * i], xs[2 * i + 1]), a, re, im);
                                sum += re * re + im * im;
                            }
                        }
                        sum /= den;
                        for (int kk = table[lane]; kk < table[lane + 1]; kk++)
                            if (kk - kx < MAXM) w.sumA[e][kk - kx] = sum;
                    }
                }
            }
            wave_sync();
            if (in_sbr) {
#pragma unroll
                for (int e = 0; e < MAXE; e++)
                    if (e < num_env) e_curr[e] = w.sumA[e][m];
            }
            wave_sync();
        }

        HSTAMP(7);
        // ---- sbr_gain_calc (:1552-1605) ----
        // elementwise parts per lane, limiter-band sums by one lane per (envelope, band)
        const int lim = in_sbr ? (int)h.map_lim[k] : 0xff;
        const bool limited = lim != 0xff;
        const float limgain = h.bs_limiter_gains == 0 ? 0.70795f :
                              h.bs_limiter_gains == 1 ? 1.0f :
                              h.bs_limiter_gains == 2 ? 1.41254f : 10000000000.0f;
        const int n_lim = h.n_lim;
#pragma unroll
        for (int e = 0; e < MAXE; e++) {
            if (e < num_env && limited) {
                const int delta = !((e == c.e_a[1]) || (e == c.e_a[0]));
                const float eo = e_orig[e], qm = q_map[e], ec = e_curr[e];
                const float temp = eo / (1.0f + qm);
                q_m[e] = sqrtf(temp * qm);
                s_m[e] = sqrtf(temp * (float)sidx[e]);
                if (!((smap >> e) & 1))
                    gain[e] = sqrtf(eo / ((1.0f + ec) * (1.0f + qm * (float)delta)));
                else
                    gain[e] = sqrtf(eo * qm / ((1.0f + ec) * (1.0f + qm)));
                w.sumA[e][m] = eo;
                w.sumB[e][m] = ec;
            }
        }
        wave_sync();
        for (int t = lane; t < num_env * n_lim; t += WAVE) {
            const int e = t / n_lim, kk = t - e * n_lim;
            const int ma = h.f_tablelim[kk] - kx, mb = h.f_tablelim[kk + 1] - kx;
            float sum0 = 0.0f, sum1 = 0.0f;
            for (int mm = ma; mm < mb; mm++) {
                sum0 += w.sumA[e][mm];
                sum1 += w.sumB[e][mm];
            }
            float gain_max = limgain * sqrtf((1.1920928955078125e-7f + sum0) / (1.1920928955078125e-7f + sum1));
            gain_max = FFMIN_(100000.0f, gain_max);
            w.bandv[e][kk] = gain_max;
        }
        wave_sync();
#pragma unroll
        for (int e = 0; e < MAXE; e++) {
            if (e < num_env && limited) {
                const int delta = !((e == c.e_a[1]) || (e == c.e_a[0]));
                const float gain_max = w.bandv[e][lim];
                const float q_m_max = q_m[e] * gain_max / gain[e];
                q_m[e]  = FFMIN_(q_m[e], q_m_max);
                gain[e] = FFMIN_(gain[e], gain_max);
                // term of the second sum[1] (:1590-1594); sumA still holds e_origmapped
                w.sumB[e][m] = e_curr[e] * gain[e] * gain[e]
                               + s_m[e] * s_m[e]
                               + (float)(delta && !s_m[e]) * q_m[e] * q_m[e];
            }
        }
        wave_sync();
        for (int t = lane; t < num_env * n_lim; t += WAVE) {
            const int e = t / n_lim, kk = t - e * n_lim;
            const int ma = h.f_tablelim[kk] - kx, mb = h.f_tablelim[kk + 1] - kx;
            float sum0 = 0.0f, sum1 = 0.0f;
            for (int mm = ma; mm < mb; mm++) {
                sum0 += w.sumA[e][mm];
                sum1 += w.sumB[e][mm];
            }
            float gain_boost = sqrtf((1.1920928955078125e-7f + sum0) / (1.1920928955078125e-7f + sum1));
            // FFMIN(1.584893192, gain_boost) is evaluated in double (:1597)
            gain_boost = (float)(1.584893192 > (double)gain_boost ? (double)gain_boost : 1.584893192);
            w.bandv[e][kk] = gain_boost;
        }
        wave_sync();
#pragma unroll
        for (int e = 0; e < MAXE; e++) {
            if (e < num_env && limited) {
                const float gain_boost = w.bandv[e][lim];
                gain[e] *= gain_boost;
                q_m[e]  *= gain_boost;
                s_m[e]  *= gain_boost;
            }
        }
        // history rows for the smoothing filter (:1630-1639)
        if (reset) {
#pragma unroll
            for (int j = 0; j < 4; j++) { ghist[j] = gain[0]; qhist[j] = q_m[0]; }
        }
    }

    HSTAMP(8);
    // ---- sbr_hf_assemble (:1608-1714) fused with sbr_x_gen (:1412-1446) ----
    const int t_old = c.t_env_num_env_old;
    const int i_Temp = 2 * t_old - 32 > 0 ? 2 * t_old - 32 : 0;
    const float *ytail_in = st_in + HEAAC_SBR_YTAIL;
    float *ytail_out = st_out + HEAAC_SBR_YTAIL;
    float *X0 = g_X, *X1 = g_X + 38 * 64;
    {
        const bool hf = start && in_sbr;
        const float hs[5] = { 0.33333333333333f, 0.30150283239582f, 0.21816949906249f,
                              0.11516383427084f, 0.03183050093751f };
        const int phi_sign0 = (1 - 2 * (kx & 1)) * ((m & 1) ? -1 : 1);
        // g_temp / q_temp rows r = slot + h_SL kept as a ring of 5 (position r % 5);
        // rows 2 t0 .. 2 t0 + 3 hold the history, row slot + 4 the slot's own gain
        float gr[5] = { 0, 0, 0, 0, 0 }, qr[5] = { 0, 0, 0, 0, 0 };
        // current envelope (uniform): advanced at even slots = 2 * t_env[e + 1]
        int e = 0, next_border = 2 * c.t_env[1];
        float g_e = gain[0], q_e = q_m[0], s_e = s_m[0];
        bool plain = (0 == c.e_a[0]) || (0 == c.e_a[1]);
        // sliding window of X_low of the source row
        float2 x2 = make_float2(xrow[0], xrow[1]), x1 = make_float2(xrow[2], xrow[3]);

#pragma unroll
        for (int i = 0; i < 38; i++) {
            const float2 x0 = make_float2(xrow[2 * (i + ENV_ADJ)], xrow[2 * (i + ENV_ADJ) + 1]);
            if ((i & 1) == 0 && i <= 6 && i == 2 * t0 && h_SL) {
                // seed the ring with the four history rows (:1630-1639)
#pragma unroll
                for (int j = 0; j < 4; j++) { gr[(i + j) % 5] = ghist[j]; qr[(i + j) % 5] = qhist[j]; }
            }
            if ((i & 1) == 0 && i > 0 && i == next_border && e + 1 < num_env) {
                e++;
                next_border = 2 * c.t_env[e + 1];
                // per-envelope values of this lane (static select: e is uniform)
                g_e = e == 1 ? gain[1] : e == 2 ? gain[2] : e == 3 ? gain[3] : gain[4];
                q_e = e == 1 ? q_m[1] : e == 2 ? q_m[2] : e == 3 ? q_m[3] : q_m[4];
                s_e = e == 1 ? s_m[1] : e == 2 ? s_m[2] : e == 3 ? s_m[3] : s_m[4];
                plain = (e == c.e_a[0]) || (e == c.e_a[1]);
            }
            float yr = 0.0f, yi = 0.0f;
            const bool have_y = hf && i >= 2 * t0 && i < 2 * tL;
            if (have_y) {
                float xr_, xi_;
                if (has_src) xhigh3(x2, x1, x0, kc, xr_, xi_); else { xr_ = 0.0f; xi_ = 0.0f; }
                gr[(i + 4) % 5] = g_e;
                qr[(i + 4) % 5] = q_e;
                float g_filt;
                if (h_SL && !plain) {
                    g_filt = 0.0f;
#pragma unroll
                    for (int j = 0; j < 5; j++) g_filt += gr[(i + 4 - j) % 5] * hs[j];
                } else {
                    g_filt = h_SL ? g_e : g_e;       // g_temp[i + h_SL][m] = this slot's gain
                }
                yr = xr_ * g_filt;
                yi = xi_ * g_filt;
                const int slot = i - 2 * t0;
                const int isine = (idxsine + slot) & 3;
                const int phi_re = isine == 0 ? 1 : isine == 2 ? -1 : 0;
                const int phi_im = isine == 1 ? 1 : isine == 3 ? -1 : 0;
                if (!plain) {
                    if (s_e) {
                        yr += s_e * (float)phi_re;
                        yi += s_e * (float)(phi_im * phi_sign0);
                    } else {
                        float q_filt;
                        if (h_SL) {
                            q_filt = 0.0f;
#pragma unroll
                            for (int j = 0; j < 5; j++) q_filt += qr[(i + 4 - j) % 5] * hs[j];
                        } else {
                            q_filt = q_e;              // q_temp[i][m], h_SL == 0
                        }
                        const unsigned in = (idxnoise + (unsigned)slot * m_max + m + 1) & 0x1ff;
                        yr += q_filt * g_noise[2 * in];
                        yi += q_filt * g_noise[2 * in + 1];
                    }
                } else {
                    yr += s_e * (float)phi_re;
                    yi += s_e * (float)(phi_im * phi_sign0);
                }
            }
            // ytail: Y[1][32..37]
            if (i >= 32) {
                const int o = ((i - 32) * 64 + k) * 2;
                if (have_y) { ytail_out[o] = yr; ytail_out[o + 1] = yi; }
                else if (ytail_out != ytail_in) { ytail_out[o] = ytail_in[o]; ytail_out[o + 1] = ytail_in[o + 1]; }
            }
            // x_gen
            float xo_r = 0.0f, xo_i = 0.0f;
            if (i < 6 && i < i_Temp) {
                if (k < kx_old) {
                    if (k < 32) { xo_r = w.xlow[k * XL_STRIDE + 2 * (i + ENV_ADJ)]; xo_i = w.xlow[k * XL_STRIDE + 2 * (i + ENV_ADJ) + 1]; }
                } else if (k < kx_old + m_old) {
                    xo_r = ytail_in[(i * 64 + k) * 2]; xo_i = ytail_in[(i * 64 + k) * 2 + 1];
                }
            } else {
                if (k < kx) {
                    if (k < 32) { xo_r = x0.x; xo_i = x0.y; }
                } else if (k < kx + m_max && i < 32) {
                    xo_r = yr; xo_i = yi;
                }
            }
            X0[i * 64 + k] = xo_r;
            X1[i * 64 + k] = xo_i;
            x2 = x1; x1 = x0;
        }
    }

    HSTAMP(9);
    // ---- remaining state ----
    if (start) {
        if (lane < 5) st_out[HEAAC_SBR_BW + lane] = w.bw[lane];
        if (lane == 0) {
            const unsigned slots = 2 * (tL - t0);
            st_out[HEAAC_SBR_IDXNOISE] = __uint_as_float((idxnoise + slots * m_max) & 0x1ff);
            st_out[HEAAC_SBR_IDXSINE]  = __uint_as_float((idxsine + slots) & 3);
        }
        // s_indexmapped[0] <- s_indexmapped[bs_num_env]  (bytes, one per band m < 48)
        {
            int v = 0;
#pragma unroll
            for (int e = 0; e < MAXE; e++) if (e == num_env - 1) v = sidx[e];
            // gather the byte of band m = lane (not k): shuffle from lane kx + m
            const int src_lane = kx + lane;
            const int byte = __shfl(v, src_lane < 64 ? src_lane : 0);
            const int valid = lane < MAXM && src_lane < 64 && lane < m_max;
            const int b0 = valid ? (byte & 0xff) : 0;
            // pack 4 bytes per dword via shuffles
            const int p0 = __shfl(b0, (lane & 15) * 4 + 0 < 64 ? (lane & 15) * 4 + 0 : 0);
            const int p1 = __shfl(b0, (lane & 15) * 4 + 1 < 64 ? (lane & 15) * 4 + 1 : 0);
            const int p2 = __shfl(b0, (lane & 15) * 4 + 2 < 64 ? (lane & 15) * 4 + 2 : 0);
            const int p3 = __shfl(b0, (lane & 15) * 4 + 3 < 64 ? (lane & 15) * 4 + 3 : 0);
            if (lane < 12)
                reinterpret_cast<uint32_t *>(st_out + HEAAC_SBR_SIDX)[lane] =
                    (uint32_t)p0 | ((uint32_t)p1 << 8) | ((uint32_t)p2 << 16) | ((uint32_t)p3 << 24);
        }
        if (h_SL) {
            // rows 2 tL + j: the gains of slots 2 tL - 4 + j
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int slot = 2 * tL - 4 + j;
                int ee = 0;
                for (int q = 1; q < num_env; q++)
                    if (slot >= 2 * c.t_env[q]) ee = q;
                float g = 0.0f, q = 0.0f;
#pragma unroll
                for (int e2 = 0; e2 < MAXE; e2++) if (e2 == ee) { g = gain[e2]; q = q_m[e2]; }
                if (m >= 0 && m < MAXM) {
                    st_out[HEAAC_SBR_GTAIL + j * MAXM + m] = in_sbr ? g : 0.0f;
                    st_out[HEAAC_SBR_QTAIL + j * MAXM + m] = in_sbr ? q : 0.0f;
                }
            }
            // bands m that no lane covers (kx + m >= 64) are beyond m_max: zero
            if (lane < MAXM && kx + lane >= 64) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    st_out[HEAAC_SBR_GTAIL + j * MAXM + lane] = 0.0f;
                    st_out[HEAAC_SBR_QTAIL + j * MAXM + lane] = 0.0f;
                }
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
    HSTAMP(10);
}

__global__ __launch_bounds__(HF_WAVES * WAVE)
void k_hfadj(const float *__restrict__ g_tab,
             const HeaacSbrFrame *__restrict__ g_sbr, const HeaacSbrHeader *__restrict__ g_hdr,
             const float *g_W, const float *g_state_in, float *g_state_out, int state_words,
             int ncore, int off_sbr0, float *g_X, unsigned long long n_units)
{
    __shared__ HfWave S[HF_WAVES];
    __shared__ float s_noise[1024];              // sbr_noise_table, staged once per workgroup
    wg_copy_f4(s_noise, g_tab + TB_NOISE, 1024);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE), lane = threadIdx.x % WAVE;
    for (unsigned long long u = (unsigned long long)blockIdx.x * HF_WAVES + wave; u < n_units;
         u += (unsigned long long)gridDim.x * HF_WAVES) {
        const unsigned long long f = u / ncore;
        const int ch = (int)(u - f * ncore);
        const int off = off_sbr0 + ch * HEAAC_ST_SBR;
        hf_channel(S[wave], s_noise, &g_sbr[f], g_hdr, ch, g_W + u * 2048,
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
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE), lane = threadIdx.x % WAVE;
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
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE), lane = threadIdx.x % WAVE;
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
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE), lane = threadIdx.x % WAVE;
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

#ifdef HF_STAMPS
extern "C" int heaac_debug_hf_stamps(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hf_stamps), sizeof(g_hf_stamps)) == hipSuccess ? 0 : -1;
}
#endif
