// k_tools.hip -- spectral tools on the dequantised spectrum, before the IMDCT:
// apply_mid_side_stereo (aacdec.c:1390-1411), apply_intensity_stereo (:1420-1451),
// apply_tns (:1698-1736) with compute_lpc_coefs (lpc.h:61-103).
//
// One wavefront owns one channel element: both spectra (8 KiB) and the side-info record
// sit in LDS.  M/S and intensity walk the scalefactor bands in the reference's order
// (band conditions are wave-uniform) with the lanes spread over a band's coefficients.
// TNS is an all-pole recursion along frequency -- serial by nature -- so one lane runs one
// (channel, window): 2 lanes for long windows, 16 for eight short ones.  Every difference
// and product is the reference's, in its order.
#include "k_common.h"
#include "kernels.h"

#define TL_WAVES 8

struct ToolsWave {
    float coef[2][1024];
    HeaacToolsFrame t;
    float lpc[16][HEAAC_TNS_MAX_ORDER];      // per (channel, window) lane
};

// lcg_random (aacdec.c:502-505) j steps ahead: x -> mulA[j] * x + addC[j]  (mod 2^32)
#define LCG_SKIP 97
struct LcgSkip { unsigned mulA[LCG_SKIP], addC[LCG_SKIP]; };

// The NOISE_BT branch of decode_spectrum_and_dequant (aacdec.c:1003-1029) for one channel.
// The generator is sequential in the reference; element k of a band is k + 1 steps ahead of
// the state at the band's start, so the lanes jump there directly.  The band energy is the
// reference's left-to-right sum (every lane forms it from LDS).
__device__ __forceinline__ unsigned tools_pns(ToolsWave &w, const LcgSkip &K, int ch, unsigned rs, int lane)
{
    const HeaacToolsIcs &ics = w.t.ch[ch].ics;
    float *coef = w.coef[ch];
    int idx = 0, base = 0;
    for (int g = 0; g < ics.num_window_groups; g++) {
        const int glen = ics.group_len[g];
        for (int i = 0; i < ics.max_sfb; i++, idx++) {
            if (w.t.ch[ch].band_type[idx] != HEAAC_NOISE_BT) continue;
            const int o = ics.swb_offset[i], len = ics.swb_offset[i + 1] - o;
            const float sf = w.t.ch[ch].sf[idx];
            for (int group = 0; group < glen; group++) {
                float *cfo = coef + base + group * 128 + o;
                for (int k = lane; k < len; k += WAVE)
                    cfo[k] = (float)(int)(K.mulA[k + 1] * rs + K.addC[k + 1]);
                rs = K.mulA[len] * rs + K.addC[len];
                wave_sync();
                float band_energy = 0.0f;
                for (int k = 0; k < len; k++) band_energy += cfo[k] * cfo[k];
                const float scale = sf / sqrtf(band_energy);
                wave_sync();
                for (int k = lane; k < len; k += WAVE) cfo[k] = cfo[k] * scale;
            }
        }
        base += glen * 128;
    }
    wave_sync();
    return rs;
}

__device__ __forceinline__ void tools_mid_side(ToolsWave &w, int lane)
{
    const HeaacToolsIcs &ics = w.t.ch[0].ics;
    int idx = 0, base = 0;
    for (int g = 0; g < ics.num_window_groups; g++) {
        const int glen = ics.group_len[g];
        for (int i = 0; i < ics.max_sfb; i++, idx++) {
            if (w.t.ms_mask[idx] && w.t.ch[0].band_type[idx] < HEAAC_NOISE_BT &&
                w.t.ch[1].band_type[idx] < HEAAC_NOISE_BT) {
                const int o = ics.swb_offset[i], len = ics.swb_offset[i + 1] - o;
                for (int e = lane; e < glen * len; e += WAVE) {
                    const int group = e / len, k = e - group * len;
                    const int p = base + group * 128 + o + k;
                    const float a = w.coef[0][p], b = w.coef[1][p];     // butterflies_float_c
                    w.coef[0][p] = a + b;
                    w.coef[1][p] = a - b;
                }
            }
        }
        base += glen * 128;
    }
}

__device__ __forceinline__ void tools_intensity(ToolsWave &w, int lane)
{
    const HeaacToolsIcs &ics = w.t.ch[1].ics;
    int idx = 0, base = 0;
    for (int g = 0; g < ics.num_window_groups; g++) {
        const int glen = ics.group_len[g];
        for (int i = 0; i < ics.max_sfb; i++, idx++) {
            const int bt = w.t.ch[1].band_type[idx];
            if (bt == HEAAC_INTENSITY_BT || bt == HEAAC_INTENSITY_BT2) {
                int c = -1 + 2 * (bt - 14);
                if (w.t.ms_present) c *= 1 - 2 * w.t.ms_mask[idx];
                const float scale = c * w.t.ch[1].sf[idx];
                const int o = ics.swb_offset[i], len = ics.swb_offset[i + 1] - o;
                for (int e = lane; e < glen * len; e += WAVE) {
                    const int group = e / len, k = e - group * len;
                    const int p = base + group * 128 + o + k;
                    w.coef[1][p] = scale * w.coef[0][p];
                }
            }
        }
        base += glen * 128;
    }
}

// flt16_round / flt16_even / flt16_trunc, aacdec.c:1247-1269 (flt16_even's `& 0x00010000U >> 16`
// parses as `& 1`: kept)
__device__ __forceinline__ float flt16_round(float pf)
{
    return __uint_as_float((__float_as_uint(pf) + 0x00008000u) & 0xFFFF0000u);
}
__device__ __forceinline__ float flt16_even(float pf)
{
    const unsigned i = __float_as_uint(pf);
    return __uint_as_float((i + 0x00007FFFu + (i & 1u)) & 0xFFFF0000u);
}
__device__ __forceinline__ float flt16_trunc(float pf)
{
    return __uint_as_float(__float_as_uint(pf) & 0xFFFF0000u);
}

// apply_prediction (aacdec.c:1302-1322) for one channel: the 672 predictors are independent, one
// per lane and pass.  predict() (:1271-1297) is restated with its mixed precision: the two
// variance updates add a float product to 0.5 * (double) -- the literal is a double there.
__device__ __forceinline__ void tools_prediction(ToolsWave &w, int ch, const HeaacPredictorState *g_in,
                                                 HeaacPredictorState *g_out, int lane)
{
    const HeaacToolsIcs &ics = w.t.ch[ch].ics;
    const HeaacPrediction &pr = w.t.ch[ch].pred;
    const bool eight = ics.num_windows == 8;
    const int limit = eight ? 0 : ics.swb_offset[pr.pred_sfb_max];        // lines [0, limit) are predicted
    const int group = pr.predictor_reset_group;
    const float sf_scale = HEAAC_SF_SCALE;
    const float a = 0.953125f, alpha = 0.90625f;
    for (int k = lane; k < HEAAC_MAX_PREDICTORS; k += WAVE) {
        HeaacPredictorState ps = g_in[k];
        if (k < limit) {
            // scalefactor band of line k (bands are at most 96 wide: walk from a coarse guess)
            int sfb = 0;
            while (ics.swb_offset[sfb + 1] <= k) sfb++;
            const bool output_enable = pr.predictor_present && pr.prediction_used[sfb];
            float coef = w.coef[ch][k];
            const float k1 = ps.var0 > 1 ? ps.cor0 * flt16_even(a / ps.var0) : 0.0f;
            const float k2 = ps.var1 > 1 ? ps.cor1 * flt16_even(a / ps.var1) : 0.0f;
            const float pv = flt16_round(k1 * ps.r0 + k2 * ps.r1);
            if (output_enable) coef += pv * sf_scale;
            const float e0 = coef / sf_scale;
            const float e1 = e0 - k1 * ps.r0;
            const float c1 = flt16_trunc(alpha * ps.cor1 + ps.r1 * e1);
            const float v1 = flt16_trunc((float)((double)(alpha * ps.var1) + 0.5 * (double)(ps.r1 * ps.r1 + e1 * e1)));
            const float c0 = flt16_trunc(alpha * ps.cor0 + ps.r0 * e0);
            const float v0 = flt16_trunc((float)((double)(alpha * ps.var0) + 0.5 * (double)(ps.r0 * ps.r0 + e0 * e0)));
            const float r1 = flt16_trunc(a * (ps.r0 - k1 * e0));
            const float r0 = flt16_trunc(a * e0);
            ps.cor0 = c0; ps.cor1 = c1; ps.var0 = v0; ps.var1 = v1; ps.r0 = r0; ps.r1 = r1;
            w.coef[ch][k] = coef;
        }
        // reset_predictor_group (:524-529) / reset_all_predictors for eight short windows
        if (eight || (group && k % 30 == group - 1)) {
            ps.cor0 = ps.cor1 = ps.r0 = ps.r1 = 0.0f;
            ps.var0 = ps.var1 = 1.0f;
        }
        g_out[k] = ps;
    }
    wave_sync();
}

// one lane = window `win` of channel `ch`
__device__ __forceinline__ void tools_tns_window(ToolsWave &w, int ch, int win, float *lpc)
{
    const HeaacTns &tns = w.t.ch[ch].tns;
    const HeaacToolsIcs &ics = w.t.ch[ch].ics;
    float *coef = w.coef[ch];
    const int mmm = ics.tns_max_bands < ics.max_sfb ? ics.tns_max_bands : ics.max_sfb;
    int bottom = ics.num_swb;
    for (int filt = 0; filt < tns.n_filt[win]; filt++) {
        const int top = bottom;
        bottom = top - tns.length[win][filt] > 0 ? top - tns.length[win][filt] : 0;
        const int order = tns.order[win][filt];
        if (order == 0) continue;
        // compute_lpc_coefs(coef, order, lpc, 0, 0, 0)
        for (int i = 0; i < order; i++) {
            const float r = -tns.coef[win][filt][i];
            lpc[i] = r;
            for (int j = 0; j < (i + 1) >> 1; j++) {
                const float f = lpc[j], b = lpc[i - 1 - j];
                lpc[j]         = f + r * b;
                lpc[i - 1 - j] = b + r * f;
            }
        }
        int start = ics.swb_offset[bottom < mmm ? bottom : mmm];
        const int end = ics.swb_offset[top < mmm ? top : mmm];
        const int size = end - start;
        if (size <= 0) continue;
        int inc = 1;
        if (tns.direction[win][filt]) { inc = -1; start = end - 1; }
        start += win * 128;
        // ar filter
        for (int m = 0; m < size; m++, start += inc) {
            float acc = coef[start];
            const int lim = m < order ? m : order;
            for (int i = 1; i <= lim; i++)
                acc -= coef[start - i * inc] * lpc[i - 1];
            coef[start] = acc;
        }
    }
}

// apply_channel_coupling (aacdec.c:1870-1898) with apply_dependent_coupling (:1813-1843) at one coupling point:
// every coupling element of the access unit (slots in ascending tag order) adds gain * its spectrum into the
// target channels its links name.  The band walk is wave-uniform, the lanes spread over a band's lines.
template <int CH>
__device__ __forceinline__ void tools_dependent_coupling(ToolsWave &w, const HeaacCceFrame *cce, const float *cce_coeffs,
                                                         int n_cce, int point, int lane)
{
    for (int e = 0; e < n_cce; e++) {
        const HeaacCceFrame &c = cce[e];
        if (!c.present || c.coupling_point != point) continue;
        const float *src0 = cce_coeffs + e * 1024;
        const int n_links = c.n_links < HEAAC_MAX_CCE_LINKS ? c.n_links : HEAAC_MAX_CCE_LINKS;
        for (int l = 0; l < n_links; l++) {
            const HeaacCceLink &k = c.link[l];
            if (k.target_ch >= CH) continue;
            float *dest = w.coef[k.target_ch];
            const float *src = src0;
            int idx = 0;
            for (int g = 0; g < c.ics.num_window_groups && g < 8; g++) {
                const int glen = c.ics.group_len[g];
                for (int i = 0; i < c.ics.max_sfb && idx < 120; i++, idx++) {
                    if (c.band_type[idx] == 0) continue;                       // ZERO_BT
                    const float gain = k.gain[idx];
                    const int o = c.ics.swb_offset[i], len = c.ics.swb_offset[i + 1] - o;
                    for (int t = lane; t < glen * len; t += WAVE) {
                        const int group = t / len, kk = o + t - group * len;
                        const int p = group * 128 + kk;
                        if (dest + p - w.coef[k.target_ch] < 1024 && src + p - src0 < 1024)
                            dest[p] += gain * src[p];
                    }
                }
                dest += glen * 128;
                src += glen * 128;
            }
            wave_sync();
        }
    }
}

template <int CH>
__global__ __launch_bounds__(TL_WAVES * WAVE)
void k_spectral_tools(float *g_coeffs, const HeaacToolsFrame *__restrict__ g_tools,
                      const int *g_rng_in, int *g_rng_out,
                      const HeaacPredictorState *g_pred_in, HeaacPredictorState *g_pred_out,
                      int stages, const HeaacCceFrame *__restrict__ g_cce, const float *__restrict__ g_cce_coeffs,
                      int n_cce, unsigned long long n)
{
    __shared__ ToolsWave S[TL_WAVES];
    __shared__ LcgSkip K;
    if (threadIdx.x == 0) {
        unsigned a = 1u, c = 0u;                 // identity, then compose one step at a time
        for (int j = 0; j < LCG_SKIP; j++) {
            K.mulA[j] = a; K.addC[j] = c;
            a = a * 1664525u;
            c = c * 1664525u + 1013904223u;
        }
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE), lane = threadIdx.x % WAVE;
    ToolsWave &w = S[wave];
    for (unsigned long long f = (unsigned long long)blockIdx.x * TL_WAVES + wave; f < n;
         f += (unsigned long long)gridDim.x * TL_WAVES) {
        float *gc = g_coeffs + f * CH * 1024;
        {
            const float4 *c4 = reinterpret_cast<const float4 *>(gc);
            float4 *d4 = reinterpret_cast<float4 *>(&w.coef[0][0]);
#pragma unroll
            for (int i = 0; i < 4 * CH; i++) d4[lane + 64 * i] = c4[lane + 64 * i];
            const uint32_t *s = reinterpret_cast<const uint32_t *>(&g_tools[f]);
            uint32_t *d = reinterpret_cast<uint32_t *>(&w.t);
            for (int i = lane; i < (int)(sizeof(HeaacToolsFrame) / 4); i += WAVE) d[i] = s[i];
        }
        wave_sync();
        const bool pre = stages & HEAAC_TOOLS_PRE, post = stages & HEAAC_TOOLS_POST;
        if (pre && g_rng_in) {
            unsigned rs = (unsigned)g_rng_in[f];
#pragma unroll
            for (int c = 0; c < CH; c++) rs = tools_pns(w, K, c, rs, lane);
            if (lane == 0) g_rng_out[f] = (int)rs;
        }
        const bool common = CH == 2 && w.t.common_window;
        if (pre && g_pred_in && !common) {      // decode_ics, aacdec.c:1381-1382
#pragma unroll
            for (int c = 0; c < CH; c++)
                tools_prediction(w, c, g_pred_in + (f * CH + c) * HEAAC_MAX_PREDICTORS,
                                 g_pred_out + (f * CH + c) * HEAAC_MAX_PREDICTORS, lane);
        }
        if (CH == 2 && pre) {
            if (w.t.common_window && w.t.ms_present) { tools_mid_side(w, lane); wave_sync(); }
            if (g_pred_in && common) {          // decode_cpe, aacdec.c:1486-1489
                for (int c = 0; c < 2; c++)
                    tools_prediction(w, c, g_pred_in + (f * 2 + c) * HEAAC_MAX_PREDICTORS,
                                     g_pred_out + (f * 2 + c) * HEAAC_MAX_PREDICTORS, lane);
            }
            tools_intensity(w, lane);
            wave_sync();
        }
        if (post) {
            if (n_cce)
                tools_dependent_coupling<CH>(w, g_cce + f * n_cce, g_cce_coeffs + f * n_cce * 1024, n_cce,
                                             HEAAC_CC_BEFORE_TNS, lane);
            // lane -> (channel, window)
            const int ch = lane >> 3, win = lane & 7;
            if (ch < CH && w.t.ch[ch].tns.present && win < w.t.ch[ch].ics.num_windows)
                tools_tns_window(w, ch, win, w.lpc[lane]);
            wave_sync();
            if (n_cce)
                tools_dependent_coupling<CH>(w, g_cce + f * n_cce, g_cce_coeffs + f * n_cce * 1024, n_cce,
                                             HEAAC_CC_BETWEEN_TNS_AND_IMDCT, lane);
        }
        wave_sync();
        {
            float4 *c4 = reinterpret_cast<float4 *>(gc);
            const float4 *d4 = reinterpret_cast<const float4 *>(&w.coef[0][0]);
#pragma unroll
            for (int i = 0; i < 4 * CH; i++) c4[lane + 64 * i] = d4[lane + 64 * i];
        }
        wave_sync();
    }
}

extern "C" int heaac_launch_spectral_tools(int channels, float *d_coeffs, const HeaacToolsFrame *d_tools,
                                           const int *d_rng_in, int *d_rng_out,
                                           const HeaacPredictorState *d_pred_in, HeaacPredictorState *d_pred_out,
                                           int stages, const HeaacCceFrame *d_cce, const float *d_cce_coeffs, int n_cce,
                                           size_t n, hipStream_t s)
{
    if (n == 0) return HEAAC_OK;
    unsigned long long g = (n + TL_WAVES - 1) / TL_WAVES;
    if (g > 256) g = 256;
    if (channels == 2)
        hipLaunchKernelGGL(k_spectral_tools<2>, dim3((unsigned)g), dim3(TL_WAVES * WAVE), 0, s, d_coeffs, d_tools,
                           d_rng_in, d_rng_out, d_pred_in, d_pred_out, stages, d_cce, d_cce_coeffs, n_cce,
                           (unsigned long long)n);
    else if (channels == 1)
        hipLaunchKernelGGL(k_spectral_tools<1>, dim3((unsigned)g), dim3(TL_WAVES * WAVE), 0, s, d_coeffs, d_tools,
                           d_rng_in, d_rng_out, d_pred_in, d_pred_out, stages, d_cce, d_cce_coeffs, n_cce,
                           (unsigned long long)n);
    else
        return HEAAC_ERR_ARG;
    return hipGetLastError() == hipSuccess ? HEAAC_OK : HEAAC_ERR_HIP;
}
