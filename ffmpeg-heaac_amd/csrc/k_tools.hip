// k_tools.hip -- spectral tools on the dequantised spectrum, before the IMDCT:
// apply_mid_side_stereo (aacdec.c:1390-1411), apply_intensity_stereo (:1420-1451),
// apply_tns (:1698-1736) with compute_lpc_coefs (lpc.h:61-103).
//
// One wavefront owns one channel element: both spectra (8 KiB) and the side-info record
// sit in LDS.  M/S and intensity walk the scalefactor bands in the reference's order
// (band conditions are wave-uniform) with the lanes spread over a band's coefficients.
// TNS is an all-pole recursion along frequency -- serial by nature -- so one lane runs one
// (channel, window, filter) with the recursion's state in its registers: up to 6 lanes for long
// windows, 16 for eight short ones.  Every difference and product is the reference's, in its order.
#include "k_common.h"
#include "kernels.h"

#define TL_WAVES 10

struct ToolsWave {
    float coef[2][1024];
    HeaacToolsFrame t;
};

// lcg_random (aacdec.c:502-505) j steps ahead: x -> mulA[j] * x + addC[j]  (mod 2^32)
#define LCG_SKIP 97
struct LcgSkip { unsigned mulA[LCG_SKIP], addC[LCG_SKIP]; };

// The NOISE_BT branch of decode_spectrum_and_dequant (aacdec.c:1003-1029) for one channel.
// The generator is sequential in the reference, but a band's start state only depends on how many numbers the bands in
// front of it draw: x -> A x + C maps compose (mod 2^32), so every lane forms the map of ITS band (idx = g * max_sfb +
// sfb: group_len * width draws for a noise band, the identity otherwise), a prefix scan over the lanes gives each
// band the state it starts from, and the bands run side by side -- each lane draws its band's numbers window by
// window, sums the energy left to right as scalarproduct_float_c does, and scales (:1016-1029).
struct Affine { unsigned a, c; };           // x -> a x + c
__device__ __forceinline__ Affine after(Affine first, Affine then) { return Affine{ then.a * first.a, then.a * first.c + then.c }; }

__device__ __forceinline__ unsigned tools_pns(ToolsWave &w, const LcgSkip &K, int ch, unsigned rs, int lane)
{
    const HeaacToolsIcs &ics = w.t.ch[ch].ics;
    float *coef = w.coef[ch];
    const int ng = ics.num_window_groups < 8 ? ics.num_window_groups : 8;
    const int nb = ng * ics.max_sfb < 128 ? ng * ics.max_sfb : 128;
    for (int first = 0; first < nb; first += WAVE) {                 // (at most two rounds: 120 bands)
        const int idx = first + lane;
        const bool mine = idx < nb && w.t.ch[ch].band_type[idx] == HEAAC_NOISE_BT;
        int g = 0, i = 0, glen = 0, o = 0, len = 0, base = 0;
        if (mine) {
            g = idx / ics.max_sfb; i = idx - g * ics.max_sfb;
            for (int q = 0; q < g; q++) base += ics.group_len[q] * 128;
            glen = ics.group_len[g];
            o = ics.swb_offset[i];
            len = ics.swb_offset[i + 1] - o;
            // (a record no band table gives must not become an address outside the channel's 1024 lines: nothing drawn)
            if (len < 0 || len >= LCG_SKIP || glen < 1 || glen > 8 || base + (glen - 1) * 128 + o + len > 1024) len = 0;
        }
        // this band's map: glen times the map of `len` draws
        Affine m = { 1u, 0u };
        const Affine one = { K.mulA[len], K.addC[len] };
        for (int q = 0; q < glen; q++) m = after(m, one);
        // inclusive scan over the lanes
        Affine p = m;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const unsigned oa = __shfl_up(p.a, off), oc = __shfl_up(p.c, off);
            if (lane >= off) p = after(Affine{ oa, oc }, p);
        }
        // the state this band starts from: the lanes in front of it, applied to the state the round starts from
        unsigned ea = __shfl_up(p.a, 1), ec = __shfl_up(p.c, 1);
        if (lane == 0) { ea = 1u; ec = 0u; }
        unsigned r = ea * rs + ec;
        const unsigned ta = __shfl(p.a, WAVE - 1), tc = __shfl(p.c, WAVE - 1);
        rs = ta * rs + tc;                                            // ... and the state the next round (or channel) starts from
        if (mine && len > 0) {
            const float sf = w.t.ch[ch].sf[idx];
            for (int group = 0; group < glen; group++) {
                float *cfo = coef + base + group * 128 + o;
                float band_energy = 0.0f;
                for (int k = 0; k < len; k++) {
                    r = r * 1664525u + 1013904223u;
                    const float v = (float)(int)r;
                    cfo[k] = v;
                    band_energy += v * v;
                }
                const float scale = sf / sqrtf(band_energy);
                for (int k = 0; k < len; k++) cfo[k] = cfo[k] * scale;
            }
        }
    }
    wave_sync();
    return rs;
}

// Where band idx = g * max_sfb + sfb of a channel's grouping lies: its first line, its width, the windows of its
// group.  A record no band table gives (lines beyond the channel's 1024) yields len = 0.
struct BandPlace { int first, len, glen; };
__device__ __forceinline__ BandPlace band_place(const HeaacToolsIcs &ics, int idx)
{
    const int g = idx / ics.max_sfb, i = idx - g * ics.max_sfb;
    int base = 0;
    for (int q = 0; q < g && q < 8; q++) base += ics.group_len[q] * 128;
    BandPlace b = { base + ics.swb_offset[i], ics.swb_offset[i + 1] - ics.swb_offset[i], g < 8 ? ics.group_len[g] : 0 };
    if (b.len < 0 || b.glen < 1 || b.glen > 8 || b.first + (b.glen - 1) * 128 + b.len > 1024) b.len = 0;
    return b;
}

// apply_mid_side_stereo (aacdec.c:1390-1411) and apply_intensity_stereo (:1420-1451) touch every band on its own:
// one lane per band (two rounds for up to 120 of them), each walking its band's lines window by window.
__device__ __forceinline__ void tools_mid_side(ToolsWave &w, int lane)
{
    const HeaacToolsIcs &ics = w.t.ch[0].ics;
    const int ng = ics.num_window_groups < 8 ? ics.num_window_groups : 8;
    const int nb = ng * ics.max_sfb < 128 ? ng * ics.max_sfb : 128;
    for (int idx = lane; idx < nb; idx += WAVE) {
        if (!(w.t.ms_mask[idx] && w.t.ch[0].band_type[idx] < HEAAC_NOISE_BT && w.t.ch[1].band_type[idx] < HEAAC_NOISE_BT)) continue;
        const BandPlace b = band_place(ics, idx);
        for (int group = 0; group < b.glen; group++)
            for (int k = 0; k < b.len; k++) {
                const int p = b.first + group * 128 + k;
                const float x = w.coef[0][p], y = w.coef[1][p];         // butterflies_float_c
                w.coef[0][p] = x + y;
                w.coef[1][p] = x - y;
            }
    }
}

__device__ __forceinline__ void tools_intensity(ToolsWave &w, int lane)
{
    const HeaacToolsIcs &ics = w.t.ch[1].ics;
    const int ng = ics.num_window_groups < 8 ? ics.num_window_groups : 8;
    const int nb = ng * ics.max_sfb < 128 ? ng * ics.max_sfb : 128;
    for (int idx = lane; idx < nb; idx += WAVE) {
        const int bt = w.t.ch[1].band_type[idx];
        if (bt != HEAAC_INTENSITY_BT && bt != HEAAC_INTENSITY_BT2) continue;
        int c = -1 + 2 * (bt - 14);
        if (w.t.ms_present) c *= 1 - 2 * w.t.ms_mask[idx];
        const float scale = c * w.t.ch[1].sf[idx];
        const BandPlace b = band_place(ics, idx);
        for (int group = 0; group < b.glen; group++)
            for (int k = 0; k < b.len; k++) {
                const int p = b.first + group * 128 + k;
                w.coef[1][p] = scale * w.coef[0][p];
            }
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
            while (sfb < 62 && ics.swb_offset[sfb + 1] <= k) sfb++;        // (sfb < 62: a record no band table gives must not walk off)
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

// The all-pole filter of one TNS filter (apply_tns, aacdec.c:1722-1733) over its `size` lines from `start` in steps of
// `inc`: every output is the input minus the last min(m, order) OUTPUTS times the LPC coefficients, subtracted one
// product at a time in the reference's order.  The recursion is serial; what the lane can do is keep it out of the
// LDS: coefficients and the last MAXORD outputs live in registers, the next input is on its way while the chain of
// the current one runs.  (Terms beyond min(m, order) are skipped, not fed zeros: x - (-0) would turn a -0 into +0.)
template <int MAXORD>
__device__ __forceinline__ void tns_ar(float *coef, int start, int inc, int size, int order, const float *tcoef)
{
    // compute_lpc_coefs(coef, order, lpc, 0, 0, 0)  (lpc.h:61-103, LPC_TYPE float, no normalisation)
    float lpc[MAXORD], h[MAXORD];
#pragma unroll
    for (int i = 0; i < MAXORD; i++) { lpc[i] = 0.0f; h[i] = 0.0f; }
#pragma unroll
    for (int i = 0; i < MAXORD; i++) {
        if (i < order) {
            const float r = -tcoef[i];
            lpc[i] = r;
#pragma unroll
            for (int j = 0; j < (i + 1) >> 1; j++) {
                const float f = lpc[j], b = lpc[i - 1 - j];
                lpc[j]         = f + r * b;
                lpc[i - 1 - j] = b + r * f;
            }
        }
    }
    float nxt = coef[start];
    for (int m = 0; m < size; m++, start += inc) {
        float acc = nxt;
        if (m + 1 < size) nxt = coef[start + inc];
        const int lim = m < order ? m : order;
#pragma unroll
        for (int i = 1; i <= MAXORD; i++)
            if (i <= lim) acc -= h[i - 1] * lpc[i - 1];
        coef[start] = acc;
#pragma unroll
        for (int i = MAXORD - 1; i > 0; i--) h[i] = h[i - 1];
        h[0] = acc;
    }
}

// The same recursion for 64 different (frame, channel) units at once, one per lane, straight on the spectra in global
// memory: this is what the stand-alone TNS pass runs (k_tns below).  `on`: this lane has a filter in this round;
// rounds are as long as the wave's longest filter.
template <int MAXORD>
__device__ __forceinline__ void tns_ar_lanes(float *coef, bool on, int start, int inc, int size, int order, const float *tcoef,
                                             int max_size)
{
    float lpc[MAXORD], h[MAXORD];
#pragma unroll
    for (int i = 0; i < MAXORD; i++) { lpc[i] = 0.0f; h[i] = 0.0f; }
#pragma unroll
    for (int i = 0; i < MAXORD; i++) {
        if (on && i < order) {
            const float r = -tcoef[i];
            lpc[i] = r;
#pragma unroll
            for (int j = 0; j < (i + 1) >> 1; j++) {
                const float f = lpc[j], b = lpc[i - 1 - j];
                lpc[j]         = f + r * b;
                lpc[i - 1 - j] = b + r * f;
            }
        }
    }
    float nxt = on ? coef[start] : 0.0f;
    for (int m = 0; m < max_size; m++, start += inc) {
        const bool here = on && m < size;
        float acc = nxt;
        if (on && m + 1 < size) nxt = coef[start + inc];
        const int lim = m < order ? m : order;
#pragma unroll
        for (int i = 1; i <= MAXORD; i++)
            if (i <= lim) acc -= h[i - 1] * lpc[i - 1];
        if (here) coef[start] = acc;
#pragma unroll
        for (int i = MAXORD - 1; i > 0; i--) h[i] = h[i - 1];
        h[0] = acc;
    }
}

__device__ __forceinline__ int wave_max(int v)
{
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        const int o = __shfl_xor(v, off);
        v = o > v ? o : v;
    }
    return v;
}

// apply_tns (aacdec.c:1698-1736) as a pass of its own: one lane = one of the (up to three) filters a window of one
// channel of one frame can have -- the filters of a window work on disjoint line ranges, each from its `top` down over
// `length` bands, the next one below it (:1707-1712) -- and every window round is wave-uniform (lanes without a filter
// there sit it out).  One frame per wave leaves 58 to 63 lanes idle for the whole recursion, which was most of what the
// spectral tools cost (profiles/r04_experiments.md E8).
#define TNS_FILTERS 3              // n_filt is two bits for a long window (0 .. 3), one for a short one
template <int CH>
__global__ __launch_bounds__(256)
void k_tns(float *g_coeffs, const HeaacToolsFrame *__restrict__ g_tools, unsigned long long n)
{
    const unsigned long long u = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = u < n * CH * TNS_FILTERS;
    const unsigned long long f = live ? u / (CH * TNS_FILTERS) : 0;
    const int r = live ? (int)(u - f * (CH * TNS_FILTERS)) : 0;
    const int ch = r / TNS_FILTERS, filt = r - ch * TNS_FILTERS;
    const HeaacToolsChannel &t = g_tools[f].ch[ch];
    const HeaacTns &tns = t.tns;
    const HeaacToolsIcs &ics = t.ics;
    const bool has = live && tns.present;
    if (wave_max(has) == 0) return;
    float *coef = g_coeffs + (f * CH + ch) * 1024;
    const int num_windows = has ? (ics.num_windows == 8 ? 8 : 1) : 0;
    const int mmm = ics.tns_max_bands < ics.max_sfb ? ics.tns_max_bands : ics.max_sfb;
    const int windows = wave_max(num_windows);
    for (int win = 0; win < windows; win++) {
        bool on = win < num_windows && filt < tns.n_filt[win];
        int order = 0, start = 0, size = 0, inc = 1;
        if (on) {
            int bottom = ics.num_swb, top = bottom;
            for (int q = 0; q <= filt; q++) {
                top = bottom;
                bottom = top - tns.length[win][q] > 0 ? top - tns.length[win][q] : 0;
            }
            order = tns.order[win][filt];
            if (order > HEAAC_TNS_MAX_ORDER) order = HEAAC_TNS_MAX_ORDER;
            start = ics.swb_offset[bottom < mmm ? bottom : mmm];
            const int end = ics.swb_offset[top < mmm ? top : mmm];
            size = end - start;
            if (tns.direction[win][filt]) { inc = -1; start = end - 1; }
            start += win * 128;
            // (a record no band table gives must not become an address outside the channel's 1024 lines)
            if (order == 0 || size <= 0 || start < 0 || start >= 1024 || start + inc * (size - 1) < 0 || start + inc * (size - 1) >= 1024) {
                on = false; size = 0; order = 0;
            }
        }
        const int max_size = wave_max(size), max_order = wave_max(order);
        if (max_size == 0) continue;
        const float *tc = tns.coef[win][filt];
        if (max_order <= 7)       tns_ar_lanes<7>(coef, on, start, inc, size, order, tc, max_size);
        else if (max_order <= 12) tns_ar_lanes<12>(coef, on, start, inc, size, order, tc, max_size);
        else                      tns_ar_lanes<HEAAC_TNS_MAX_ORDER>(coef, on, start, inc, size, order, tc, max_size);
    }
}

// one lane = filter `filt` of window `win` of channel `ch`: the filters of a window work on disjoint line ranges
// (each from its `top` down over `length` bands, the next one below it, :1707-1712), so they run side by side
__device__ __forceinline__ void tools_tns_filter(ToolsWave &w, int ch, int win, int filt)
{
    const HeaacTns &tns = w.t.ch[ch].tns;
    const HeaacToolsIcs &ics = w.t.ch[ch].ics;
    const int mmm = ics.tns_max_bands < ics.max_sfb ? ics.tns_max_bands : ics.max_sfb;
    int bottom = ics.num_swb, top = bottom;
    for (int f = 0; f <= filt; f++) {
        top = bottom;
        bottom = top - tns.length[win][f] > 0 ? top - tns.length[win][f] : 0;
    }
    const int order = tns.order[win][filt];
    if (order == 0) return;
    int start = ics.swb_offset[bottom < mmm ? bottom : mmm];
    const int end = ics.swb_offset[top < mmm ? top : mmm];
    const int size = end - start;
    if (size <= 0) return;
    int inc = 1;
    if (tns.direction[win][filt]) { inc = -1; start = end - 1; }
    start += win * 128;
    // (a record no band table gives must not become an address outside the channel's 1024 lines)
    if (start < 0 || start >= 1024 || start + inc * (size - 1) < 0 || start + inc * (size - 1) >= 1024) return;
    const float *tc = tns.coef[win][filt];
    if (order <= 7)       tns_ar<7>(w.coef[ch], start, inc, size, order, tc);
    else if (order <= 12) tns_ar<12>(w.coef[ch], start, inc, size, order, tc);
    else                  tns_ar<HEAAC_TNS_MAX_ORDER>(w.coef[ch], start, inc, size, order < HEAAC_TNS_MAX_ORDER ? order : HEAAC_TNS_MAX_ORDER, tc);
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
            // lane -> (channel, window, filter)
            const int ch = lane >> 5, win = (lane >> 2) & 7, filt = lane & 3;
            if (ch < CH && w.t.ch[ch].tns.present && win < w.t.ch[ch].ics.num_windows && filt < w.t.ch[ch].tns.n_filt[win])
                tools_tns_filter(w, ch, win, filt);
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
    if (channels != 1 && channels != 2) return HEAAC_ERR_ARG;
    unsigned long long g = (n + TL_WAVES - 1) / TL_WAVES;
    if (g > 256) g = 256;
    // Without coupling elements the second half is TNS alone: it runs as a pass of its own, one lane per channel of a
    // frame (k_tns), behind the first half.  With them it stays inside the frame's wave, between the two coupling points.
    const bool tns_pass = (stages & HEAAC_TOOLS_POST) && n_cce == 0;
    const int in_wave = tns_pass ? (stages & ~HEAAC_TOOLS_POST) : stages;
    if (in_wave) {
        if (channels == 2)
            hipLaunchKernelGGL(k_spectral_tools<2>, dim3((unsigned)g), dim3(TL_WAVES * WAVE), 0, s, d_coeffs, d_tools,
                               d_rng_in, d_rng_out, d_pred_in, d_pred_out, in_wave, d_cce, d_cce_coeffs, n_cce,
                               (unsigned long long)n);
        else
            hipLaunchKernelGGL(k_spectral_tools<1>, dim3((unsigned)g), dim3(TL_WAVES * WAVE), 0, s, d_coeffs, d_tools,
                               d_rng_in, d_rng_out, d_pred_in, d_pred_out, in_wave, d_cce, d_cce_coeffs, n_cce,
                               (unsigned long long)n);
    }
    if (tns_pass) {
        const unsigned long long units = (unsigned long long)n * channels * TNS_FILTERS, blocks = (units + 255) / 256;
        if (blocks > 0x7fffffffull) return HEAAC_ERR_ARG;
        if (channels == 2) hipLaunchKernelGGL(k_tns<2>, dim3((unsigned)blocks), dim3(256), 0, s, d_coeffs, d_tools, (unsigned long long)n);
        else               hipLaunchKernelGGL(k_tns<1>, dim3((unsigned)blocks), dim3(256), 0, s, d_coeffs, d_tools, (unsigned long long)n);
    }
    return hipGetLastError() == hipSuccess ? HEAAC_OK : HEAAC_ERR_HIP;
}
