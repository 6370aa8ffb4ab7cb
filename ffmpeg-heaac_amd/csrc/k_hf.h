// k_hf.h -- SBR high-frequency generation and envelope adjustment for one channel by one
// wavefront (sbr_lf_gen .. sbr_x_gen, aacsbr.c:1337-1714).  Shared by k_hfadj (k_he.hip)
// and the fused HF + Parametric Stereo kernel (k_ps.hip).
#pragma once
#include "k_common.h"
#include "heaac_dsp.h"

#define ENV_ADJ 2          // ENVELOPE_ADJUSTMENT_OFFSET, aacsbr.c:39

#ifdef HEAAC_STAMPS
// Phase timeline (-DHEAAC_STAMPS diagnostic builds only): wave 0 of every 8th workgroup accumulates the cycles
// between consecutive stamps over all the frames it processes; slot 31 = time between frames.
static __device__ unsigned long long g_tl_acc[32], g_tl_last[256], g_tl_cnt;
#define TL_STAMP(i, is_first) do { wave_sync(); \
    if (lane == 0 && (blockIdx.x & 7) == 7 && threadIdx.x < 64) { \
        const unsigned long long t_ = __builtin_readcyclecounter(), l_ = g_tl_last[blockIdx.x]; \
        if (is_first) { if (l_) atomicAdd(&g_tl_acc[31], t_ - l_); atomicAdd(&g_tl_cnt, 1ull); } \
        else atomicAdd(&g_tl_acc[i], t_ - l_); \
        g_tl_last[blockIdx.x] = t_; } } while (0)
#define HSTAMP(i) TL_STAMP(i, (i) == 0)
#else
#define HSTAMP(i) do {} while (0)
#endif

// exp2f(twice / 2.0f) for integer `twice`: exact powers of two, or sqrt(2)
// (0x3FB504F3, what glibc's exp2f(0.5f) returns) times a power of two.
// sbr_dequant's arguments are always multiples of 0.5 (aacsbr.c:1099-1125).
__device__ __forceinline__ float exp2_half_normal(int twice)
{
    const int e = twice >> 1;
    const unsigned mant = (twice & 1) ? 0x3FB504F3u : 0x3F800000u;
    return __uint_as_float(mant + ((unsigned)e << 23));
}
__device__ __forceinline__ float exp2_half(int twice)
{
    // outside the normal range (never reached by legal scalefactors): saturate
    // like exp2f does; the denormal side is rounded once more than libm's.
    // (No recursion: a recursive helper is not inlined and every call then
    // saves and restores registers.)
    if (twice > 255) return __uint_as_float(0x7F800000u);
    if (twice < -252)
        return twice < -400 ? 0.0f : exp2_half_normal(twice + 256) * 2.938735877055719e-39f;   // 2^-128
    return exp2_half_normal(twice);
}

// ===========================================================================
// K_B  HF generation + envelope adjustment + x_gen, one wave per SBR channel
// ===========================================================================
// Lane = QMF band k (m = k - kx for the SBR range).  Everything that is per band
// and per envelope (mapped scalefactors, estimated envelope, gains) lives in that
// lane's registers; the only cross-lane steps are the limiter-band sums of
// sbr_gain_calc, which go through small LDS arrays in the reference's order.
#define HF_WAVES 11               // (10 before alpha0 / alpha1 were laid over the second limiter-sum array)
#define XL_STRIDE 81              // X_low row: 40 slots * (re,im) + 1 pad (bank spread)
#define MAXM 48                   // e_origmapped[7][48] etc. in the reference (sbr.h:165-177)
#define MAXE 5

// Per-wave LDS of the HF stage, as three blocks so that a fused kernel can lay them over
// arrays of a later stage; HfWave is the view hf_channel works through.
#define HF_XLOW_WORDS (32 * XL_STRIDE)                                   // X_low[k][i][re,im]
#define HF_AUX_WORDS  (8 + 2 * MAXE * MAXM + MAXE * 32)                  // bw, sumA, sumB (under it: alpha0/1), bandv
#define HF_REC_WORDS  ((int)((sizeof(HeaacSbrHeader) + 2 * sizeof(HeaacSbrChannel)) / 4))
struct HfWave {
    float *xlow;
    float (*alpha0)[2], (*alpha1)[2];
    float *bw;
    float (*sumA)[MAXM], (*sumB)[MAXM];           // per-band terms of the limiter-band sums
    float (*bandv)[32];                           // gain_max / gain_boost per (envelope, limiter band)
    HeaacSbrHeader &h;
    HeaacSbrChannel *c;
};
__device__ __forceinline__ HfWave hf_wave_view(float *xlow, float *aux, float *rec)
{
    // alpha0 / alpha1 (inverse filter -> the per-band constants of hf_gen, and the non-interpolating envelope estimate,
    // which broadcasts through sumA) are dead before sbr_gain_calc writes sumB: they share its first 128 words
    static_assert(MAXE * MAXM >= 128, "alpha0 / alpha1 fit under sumB");
    return HfWave{ xlow,
                   reinterpret_cast<float (*)[2]>(aux + 8 + MAXE * MAXM), reinterpret_cast<float (*)[2]>(aux + 8 + MAXE * MAXM + 64),
                   aux,
                   reinterpret_cast<float (*)[MAXM]>(aux + 8),
                   reinterpret_cast<float (*)[MAXM]>(aux + 8 + MAXE * MAXM),
                   reinterpret_cast<float (*)[32]>(aux + 8 + 2 * MAXE * MAXM),
                   *reinterpret_cast<HeaacSbrHeader *>(rec),
                   reinterpret_cast<HeaacSbrChannel *>(rec + sizeof(HeaacSbrHeader) / 4) };
}

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
// the same sums on (re, im) pairs: a * (i x) is written (-a, a) * swap(x), which the packed
// multiply takes as op_sel / neg modifiers
__device__ __forceinline__ v2f swp(v2f a) { return __builtin_shufflevector(a, a, 1, 0); }
__device__ __forceinline__ v2f xhigh3_pk(v2f x2, v2f x1, v2f x0, const float *a)
{
    v2f t = bc(a[0]) * x2;
    t = t + v2f{-a[1], a[1]} * swp(x2);
    t = t + bc(a[2]) * x1;
    t = t + v2f{-a[3], a[3]} * swp(x1);
    return t + x0;
}

// emit(i, re, im) receives X[.][i][k] of this lane's band k = lane for the slots i = 0..37
// (i is a compile-time constant at every call).
// after_params(): called once the stage's parameter loads have arrived (a fused kernel stores what it
// loaded for its later stages beside them, in the same wait).
template <class Emit, class Hook = NoHook>
__device__ __forceinline__ void hf_channel(const HfWave &w, const float *g_noise /* LDS */,
                                           const HeaacSbrFrame *g_fr, const HeaacSbrHeader *g_hdr, unsigned n_hdr,
                                           int ch, const float *g_W,
                                           const float *st_in, float *st_out, int lane_in, Emit emit,
                                           Hook after_params = Hook())
{
    // redefined opaquely so that lane-derived values are not hoisted out of the unit loop
    // (where they would sit in registers for the whole kernel and get spilled)
    const int lane = opaque(lane_in);
    HSTAMP(0);
    // ---- issue every global load up front: parameters, W, state ----
    // (clamped: an index past the table must not become an address -- heaac_dsp.h, record validation)
    const int hdr_idx = g_fr->hdr < n_hdr ? g_fr->hdr : n_hdr - 1;
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
    after_params();
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
        // Three lanes per band, one lag each: every one of the reference's five running sums is
        //   s0 += a c + b d,   s1 += a d - b c      with (a, b) = x[i], (c, d) = x[i + lag]
        // for lag 0 (real_sum0; its s1 is a b - b a = +0), lag 1 (real_sum1, imag_sum1) and lag 2 (real_sum2, imag_sum2),
        // each sum in the reference's own order; the boundary terms (:1244-1253) have the same form with i = 0 and
        // i = 38.  21 bands per pass of the wave (one pass up to k0 = 21); the lag-0 lane of a band collects the eight
        // values and solves for the two coefficients as before.
        {
            const int band_in_pass = (lane * 171) >> 9;                  // lane / 3 for lane < 128
            const int lag = lane - 3 * band_in_pass;
            const int k0 = h.k0 < 32 ? h.k0 : 32;
            for (int first = 0; first < k0; first += 21) {
                const int band = first + band_in_pass;
                const bool active = lane < 63 && band < k0;
                const float *xs = w.xlow + (active ? band : 0) * XL_STRIDE;
                const float *xl = xs + 2 * lag;
                float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
                for (int i = 1; i < 38; i++) {
                    const float a = xs[2 * i], b = xs[2 * i + 1], c = xl[2 * i], d = xl[2 * i + 1];
                    s0 += a * c + b * d;
                    s1 += a * d - b * c;
                }
                // sums with the i = 0 term (phi[.][1][.] of the reference) and with the i = 38 term (phi[.][0][.])
                const float lo0 = s0 + xs[0] * xl[0] + xs[1] * xl[1];
                const float lo1 = s1 + xs[0] * xl[1] - xs[1] * xl[0];
                const float hi0 = s0 + xs[76] * xl[76] + xs[77] * xl[77];
                const float hi1 = s1 + xs[76] * xl[77] - xs[77] * xl[76];
                // lag 0: (p210, -, p100, -)   lag 1: (p110, p111, p000, p001)   lag 2: (p010, p011, -, -)
                const int l1 = lane + 1 < 64 ? lane + 1 : 63, l2 = lane + 2 < 64 ? lane + 2 : 63;
                const float p110 = __shfl(lo0, l1), p111 = __shfl(lo1, l1), p000 = __shfl(hi0, l1), p001 = __shfl(hi1, l1);
                const float p010 = __shfl(lo0, l2), p011 = __shfl(lo1, l2);
                if (active && lag == 0) {
                    const float p210 = lo0, p100 = hi0;
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
                    w.alpha0[band][0] = a0r; w.alpha0[band][1] = a0i;
                    w.alpha1[band][0] = a1r; w.alpha1[band][1] = a1i;
                }
            }
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
                            v2f x2 = v2f{xrow[2 * (ilb - 2)], xrow[2 * (ilb - 2) + 1]};
                            v2f x1 = v2f{xrow[2 * (ilb - 1)], xrow[2 * (ilb - 1) + 1]};
                            // the next sample is read one step ahead of its use (an envelope spans an
                            // even number of slots: two steps per trip)
                            v2f xa = v2f{xrow[2 * ilb], xrow[2 * ilb + 1]};
                            v2f xb = v2f{xrow[2 * ilb + 2], xrow[2 * ilb + 3]};
                            for (int i = ilb; i < iub; i += 2) {
                                const v2f na = v2f{xrow[2 * i + 4], xrow[2 * i + 5]};
                                const v2f nb = v2f{xrow[2 * i + 6], xrow[2 * i + 7]};
                                const v2f h0 = xhigh3_pk(x2, x1, xa, kc);
                                sum += h0.x * h0.x + h0.y * h0.y;
                                const v2f h1 = xhigh3_pk(x1, xa, xb, kc);
                                sum += h1.x * h1.x + h1.y * h1.y;
                                x2 = xa; x1 = xb; xa = na; xb = nb;
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
            // (four terms' reads in flight per trip; a term past the band adds +0.0f to a sum of non-negative energies,
            // which changes no bit of it, so the order of the reference's additions is kept)
            for (int mm = ma; mm < mb; mm += 4) {
                float ta[4], tb[4];
#pragma unroll
                for (int j = 0; j < 4; j++) { ta[j] = w.sumA[e][mm + j]; tb[j] = w.sumB[e][mm + j]; }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    sum0 += mm + j < mb ? ta[j] : 0.0f;
                    sum1 += mm + j < mb ? tb[j] : 0.0f;
                }
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
            // (four terms' reads in flight per trip; a term past the band adds +0.0f to a sum of non-negative energies,
            // which changes no bit of it, so the order of the reference's additions is kept)
            for (int mm = ma; mm < mb; mm += 4) {
                float ta[4], tb[4];
#pragma unroll
                for (int j = 0; j < 4; j++) { ta[j] = w.sumA[e][mm + j]; tb[j] = w.sumB[e][mm + j]; }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    sum0 += mm + j < mb ? ta[j] : 0.0f;
                    sum1 += mm + j < mb ? tb[j] : 0.0f;
                }
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
    {
        const bool hf = start && in_sbr;
        const float hs[5] = { 0.33333333333333f, 0.30150283239582f, 0.21816949906249f,
                              0.11516383427084f, 0.03183050093751f };
        const int phi_sign0 = (1 - 2 * (kx & 1)) * ((m & 1) ? -1 : 1);
        // g_temp / q_temp rows r = slot + h_SL kept as a ring of 5 (position r % 5);
        // rows 2 t0 .. 2 t0 + 3 hold the history, row slot + 4 the slot's own gain
        v2f gq[5];                                   // (g_temp, q_temp) rows of this band
#pragma unroll
        for (int j = 0; j < 5; j++) gq[j] = v2f{0.0f, 0.0f};
        // current envelope (uniform): advanced at even slots = 2 * t_env[e + 1]
        int e = 0, next_border = 2 * c.t_env[1];
        float g_e = gain[0], q_e = q_m[0], s_e = s_m[0];
        bool plain = (0 == c.e_a[0]) || (0 == c.e_a[1]);
        // sliding window of X_low of the source row
        v2f x2 = v2f{xrow[0], xrow[1]}, x1 = v2f{xrow[2], xrow[3]};

        // Slots are walked in groups of four: the LDS reads of a group (X_low samples, noise
        // table entries) are issued together ahead of the group's (uniform) branches, so the
        // loop pays one LDS latency per group instead of several per slot.  The Y tail and
        // X_low samples sbr_x_gen takes for the first slots (i < i_Temp <= 6) are fetched
        // before the loop for the same reason.
        float2 ytin[6], xlin[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            ytin[i] = make_float2(0.0f, 0.0f);
            xlin[i] = make_float2(0.0f, 0.0f);
            if (i < i_Temp) {
                if (k >= kx_old && k < kx_old + m_old)
                    ytin[i] = make_float2(ytail_in[(i * 64 + k) * 2], ytail_in[(i * 64 + k) * 2 + 1]);
                if (k < kx_old && k < 32)
                    xlin[i] = make_float2(w.xlow[k * XL_STRIDE + 2 * (i + ENV_ADJ)], w.xlow[k * XL_STRIDE + 2 * (i + ENV_ADJ) + 1]);
            }
        }
        const unsigned noise0 = idxnoise + (unsigned)(m + 1) - (unsigned)(2 * t0) * (unsigned)m_max;
        // sinusoid phase of the first slot with output (i = 2 t0: slot 0), and the step to the next
        v2f ph_cur;
        {
            const int isine = idxsine & 3;
            const int phi_re = isine == 0 ? 1 : isine == 2 ? -1 : 0;
            const int phi_im = isine == 1 ? 1 : isine == 3 ? -1 : 0;
            ph_cur = v2f{(float)phi_re, (float)(phi_im * phi_sign0)};
        }
        const v2f ph_rot = v2f{(float)-phi_sign0, (float)phi_sign0};     // (re, s im) -> (-s (s im), s re)
        v2f xq[4], nq[4];
#pragma unroll
        for (int i = 0; i < 38; i++) {
            if ((i & 3) == 0) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (i + j < 38) {
                        xq[j] = v2f{xrow[2 * (i + j + ENV_ADJ)], xrow[2 * (i + j + ENV_ADJ) + 1]};
                        const unsigned in = (noise0 + (unsigned)(i + j) * (unsigned)m_max) & 0x1ff;
                        nq[j] = v2f{g_noise[2 * in], g_noise[2 * in + 1]};
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            const v2f x0 = xq[i & 3];
            if ((i & 1) == 0 && i <= 6 && i == 2 * t0 && h_SL) {
                // seed the ring with the four history rows (:1630-1639)
#pragma unroll
                for (int j = 0; j < 4; j++) gq[(i + j) % 5] = v2f{ghist[j], qhist[j]};
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
            // Lane-dependent conditions (band inside the SBR range, patch source present,
            // sinusoid present) select values instead of branching: a divergent branch costs
            // several scalar exec-mask instructions per slot, the selects one VALU each.
            v2f Y = v2f{0.0f, 0.0f};
            const bool in_time = i >= 2 * t0 && i < 2 * tL;           // uniform
            const bool have_y = hf && in_time;
            if (in_time) {
                v2f xh = xhigh3_pk(x2, x1, x0, kc);
                xh = has_src ? xh : v2f{0.0f, 0.0f};
                gq[(i + 4) % 5] = v2f{g_e, q_e};
                // (g_filt, q_filt): this slot's values, or the 5-tap smoothing of both rows at once
                v2f f = v2f{g_e, q_e};
                if (h_SL && !plain) {
                    v2f a = v2f{0.0f, 0.0f};
#pragma unroll
                    for (int j = 0; j < 5; j++) a = a + gq[(i + 4 - j) % 5] * bc(hs[j]);
                    f = a;
                }
                Y = xh * bc(f.x);
                // phi[f_indexsine + slot] (:1676-1698): (1,0) (0,1) (-1,0) (0,-1), the imaginary part
                // signed by phi_sign0 -- each slot's pair is the previous one times i (exact: +-1, 0)
                const v2f ph = ph_cur;
                ph_cur = ph_rot * swp(ph_cur) + v2f{0.0f, 0.0f};      // (+ 0: the zero stays +0 as (float)0 is)
                const v2f y_sine = Y + bc(s_e) * ph;
                if (!plain) {
                    // sbr_noise_table[(f_indexnoise + slot m_max + m + 1) & 0x1ff] where no sinusoid sits
                    const v2f y_noise = Y + bc(f.y) * nq[i & 3];
                    Y = s_e != 0.0f ? y_sine : y_noise;
                } else {
                    Y = y_sine;
                }
                Y = hf ? Y : v2f{0.0f, 0.0f};
            }
            const float yr = Y.x, yi = Y.y;
            // ytail: Y[1][32..37]
            if (i >= 32) {
                const int o = ((i - 32) * 64 + k) * 2;
                if (have_y) { ytail_out[o] = yr; ytail_out[o + 1] = yi; }
                else if (ytail_out != ytail_in) { ytail_out[o] = ytail_in[o]; ytail_out[o + 1] = ytail_in[o + 1]; }
            }
            // x_gen
            float xo_r, xo_i;
            if (i < 6 && i < i_Temp) {
                const bool lo = k < kx_old, hi = !lo && k < kx_old + m_old;
                xo_r = lo ? xlin[i < 6 ? i : 0].x : hi ? ytin[i < 6 ? i : 0].x : 0.0f;     // xlin is 0 for k >= 32
                xo_i = lo ? xlin[i < 6 ? i : 0].y : hi ? ytin[i < 6 ? i : 0].y : 0.0f;
            } else {
                const bool lo = k < kx, hi = !lo && k < kx + m_max && i < 32;
                const bool lo32 = lo && k < 32;
                xo_r = lo32 ? x0.x : hi ? yr : 0.0f;
                xo_i = lo32 ? x0.y : hi ? yi : 0.0f;
            }
            emit(i, xo_r, xo_i);
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

