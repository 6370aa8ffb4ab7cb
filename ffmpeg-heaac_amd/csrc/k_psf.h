// k_psf.h -- Parametric Stereo for one frame by one wavefront (device code shared by the
// stand-alone k_ps kernels and the fused HF + PS kernel in k_ps.hip).
#pragma once
#include "k_common.h"
#include "heaac_dsp.h"
#include "k_hf.h"

#ifdef HEAAC_STAMPS
#define STAMP(i) TL_STAMP(16 + (i), false)
#else
#define STAMP(i) do {} while (0)
#endif
#define PS_SCHED_GROUP 4       // slots between scheduling barriers of the PS slot loop
#define SUB_STRIDE 66          // one sub-subband row: 32 slots * (re,im) + 2 pad

// Table 8.48 / 8.49 of ISO/IEC 14496-3 (aacpsdata.c:145-158): hybrid band -> parameter band
struct KtoI { signed char v[91]; };
constexpr KtoI k_to_i_20_c = {{
     1,  0,  0,  1,  2,  3,  4,  5,  6,  7,  8,  9, 10, 11, 12, 13, 14, 14, 15,
    15, 15, 16, 16, 16, 16, 17, 17, 17, 17, 17, 18, 18, 18, 18, 18, 18, 18, 18,
    18, 18, 18, 18, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19,
    19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19 }};
constexpr KtoI k_to_i_34_c = {{
     0,  1,  2,  3,  4,  5,  6,  6,  7,  2,  1,  0, 10, 10,  4,  5,  6,  7,  8,
     9, 10, 11, 12,  9, 14, 11, 12, 13, 14, 15, 16, 13, 16, 17, 18, 19, 20, 21,
    22, 22, 23, 23, 24, 24, 25, 25, 26, 26, 27, 27, 27, 28, 28, 28, 29, 29, 29,
    30, 30, 30, 31, 31, 31, 31, 32, 32, 32, 32, 33, 33, 33, 33, 33, 33, 33, 33,
    33, 33, 33, 33, 33, 33, 33, 33, 33, 33, 33, 33, 33, 33, 33 }};

// Members of every parameter band in ascending hybrid-band order: the order in
// which decorrelation() accumulates power[i][n] (aacps.c:673-678).
struct BandMembers {
    signed char kti[92];
    unsigned char order[92];      // hybrid bands sorted by (parameter band, k)
    unsigned char first[36];      // first[i] .. first[i+1]: members of band i
    int split;                    // bands [0,split) and [split,nr_par) hold about half the members each
    int quarter[5];               // the same in four parts: bands [quarter[j], quarter[j + 1])
    int eighth[9];                // and in eight
};
constexpr BandMembers make_members(const KtoI &t, int nr_bands, int nr_par)
{
    BandMembers m{};
    int pos = 0;
    for (int k = 0; k < nr_bands; k++) m.kti[k] = t.v[k];
    for (int i = 0; i < nr_par; i++) {
        m.first[i] = (unsigned char)pos;
        for (int k = 0; k < nr_bands; k++)
            if (t.v[k] == i) m.order[pos++] = (unsigned char)k;
    }
    m.first[nr_par] = (unsigned char)pos;
    m.split = nr_par;
    for (int i = 0; i <= nr_par; i++)
        if (2 * m.first[i] >= pos) { m.split = i; break; }
    m.quarter[0] = 0;
    m.quarter[4] = nr_par;
    for (int j = 1; j < 4; j++) {
        m.quarter[j] = nr_par;
        for (int i = m.quarter[j - 1]; i <= nr_par; i++)
            if (4 * m.first[i] >= j * pos) { m.quarter[j] = i; break; }
    }
    m.eighth[0] = 0;
    m.eighth[8] = nr_par;
    for (int j = 1; j < 8; j++) {
        m.eighth[j] = nr_par;
        for (int i = m.eighth[j - 1]; i <= nr_par; i++)
            if (8 * m.first[i] >= j * pos) { m.eighth[j] = i; break; }
    }
    return m;
}
__device__ constexpr BandMembers kMem20 = make_members(k_to_i_20_c, 71, 20);
__device__ constexpr BandMembers kMem34 = make_members(k_to_i_34_c, 91, 34);

// Per-wave LDS.  GENERAL = false: baseline PS -- frames that are and were 20-band with
// IPD/OPD off (what HE-AACv2 encoders emit) -- small enough for 8 waves per CU.  GENERAL = true: any
// layout, including 20 <-> 34 switches.
// SLIM (the twelve-wave fused kernel, k_hfps12): the baseline layout on the general layout's diet -- left mix in place,
// |s|^2 eight slots at a time, the 14-slot delay tail fetched as the slots need it -- 12.2 KB per wave all told.
template <bool GENERAL, bool SLIM_ = false>
struct PsWaveT {
    static constexpr bool SLIM = SLIM_;
    static constexpr bool INPLACE_L = GENERAL || SLIM_;      // the left mix overwrites the sub-subband row it was made from
    static constexpr int NSUB = GENERAL ? 32 : 10;
    static constexpr int NLOW = GENERAL ? 5 : 3;
    static constexpr int NB = GENERAL ? 91 : 71;
    static constexpr int NPAR = GENERAL ? 34 : 20;
    static constexpr bool IS_GENERAL = GENERAL;
    static constexpr int NH = GENERAL ? 8 : 4;      // H rows kept: re+im, or re only (IPD/OPD off)
    static constexpr int PNS = NB + 1;              // |s|^2 row stride (72 / 92 floats)
    // The general layout saves LDS for a fifth wave per CU (round 3): the left mix of a sub-subband overwrites the
    // sub-subband signal it was made from (one more row: the scratch row of the lanes that are not sub-subbands), and
    // |s|^2 is formed for sixteen slots at a time.
    static constexpr int SUBROWS = INPLACE_L ? NSUB + 1 : NSUB;
    static constexpr int PN_SLOTS = SLIM_ ? 8 : GENERAL ? 16 : 32;
    static constexpr int MIXROWS = INPLACE_L ? (NSUB + 1) : 2 * (NSUB + 1);     // rows of mixed output kept in the scratch
    // scratch shared by |s|^2 (until the band powers are formed) and the mixed
    // sub-subband outputs (written afterwards): max of the two, in floats
    static constexpr int SCR0 = (PN_SLOTS * PNS > MIXROWS * SUB_STRIDE) ? PN_SLOTS * PNS : MIXROWS * SUB_STRIDE;
    // (the slim kernel lays its HF stage's limiter sums and per-envelope gains under the scratch: 8 + 5 * 3 * 64 words)
    static constexpr int SCR = SLIM_ && SCR0 < 968 ? 968 : SCR0;
    // Views of separate __shared__ arrays (distinct objects for the alias analysis).
    HeaacPsFrame &p;
    float (*inb)[44][2];               // [NLOW] hybrid analysis input: 6 history + 38 current slots
    float *pn;                         // [32][PNS] |s|^2 per slot and hybrid band
    float (*sub)[SUB_STRIDE];          // [NSUB] sub-subband signals s[ks][n] (re,im interleaved)
    float (*subL)[SUB_STRIDE];         // [NSUB + 1] mixed sub-subband outputs (alias pn's memory)
    float (*subR)[SUB_STRIDE];
    float (*pw)[33];                   // [NPAR] band power, then transient gain
    float (*Hs)[NH][NPAR];             // [6]    H11,H12,H21,H22 (re[,im]) rows per envelope border
    signed char (*iid_m)[NPAR], (*icc_m)[NPAR], (*ipd_m)[NPAR], (*opd_m)[NPAR];   // [5]
    const float *hybF, *hybG;          // 20-band hybrid filters f20_0_8[8][7][2] and g1_Q2[7]: LDS copies or the table blob
};

// map_idx_* (aacps.c:461-643) as a gather: mapped value of band b.
__device__ __forceinline__ int remap_idx(const signed char *par, int num_par, int to34, int b)
{
    if (to34) {
        if (num_par == 20 || num_par == 11) {
            // map_idx_20_to_34
            const signed char src[34] = { 0, -1, 1, 2, -2, 3, 4, 4, 5, 5, 6, 7, 8, 8, 9, 9, 10,
                                          11, 12, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18, 18, 18, 19, 19 };
            const int s = src[b];
            if (s == -1) return (par[0] + par[1]) / 2;
            if (s == -2) return (par[2] + par[3]) / 2;
            return par[s];
        }
        if (num_par == 10 || num_par == 5) {
            const signed char src[34] = { 0, 0, 0, 1, 1, 1, 2, 2, 2, 2, 3, 3, 4, 4, 4, 4, 5, 5, 6, 6,
                                          7, 7, 7, 7, 8, 8, 8, 8, 9, 9, 9, 9, 9, 9 };
            if (num_par == 5 && b >= 16) return 0;            // full == 0: par_mapped[16] = 0
            return par[src[b]];
        }
        return par[b];
    }
    if (num_par == 34 || num_par == 17) {
        // map_idx_34_to_20
        switch (b) {
        case 0:  return (2 * par[0] + par[1]) / 3;
        case 1:  return (par[1] + 2 * par[2]) / 3;
        case 2:  return (2 * par[3] + par[4]) / 3;
        case 3:  return (par[4] + 2 * par[5]) / 3;
        case 4:  return (par[6] + par[7]) / 2;
        case 5:  return (par[8] + par[9]) / 2;
        case 6:  return par[10];
        case 7:  return par[11];
        case 8:  return (par[12] + par[13]) / 2;
        case 9:  return (par[14] + par[15]) / 2;
        case 10: return par[16];
        case 11: return par[17];
        case 12: return par[18];
        case 13: return par[19];
        case 14: return (par[20] + par[21]) / 2;
        case 15: return (par[22] + par[23]) / 2;
        case 16: return (par[24] + par[25]) / 2;
        case 17: return (par[26] + par[27]) / 2;
        case 18: return (par[28] + par[29] + par[30] + par[31]) / 4;
        case 19: return (par[32] + par[33]) / 2;
        }
        return 0;
    }
    if (num_par == 10 || num_par == 5) {
        if (num_par == 5 && b >= 10) return 0;                // full == 0: par_mapped[10] = 0
        return par[b >> 1];
    }
    return par[b];
}

// map_val_20_to_34 / map_val_34_to_20 (aacps.c:491-514, 598-634) as gathers.
__device__ __forceinline__ float remap_val(const float *par, int to34, int b)
{
    if (to34) {
        const signed char src[34] = { 0, -1, 1, 2, -2, 3, 4, 4, 5, 5, 6, 7, 8, 8, 9, 9, 10,
                                      11, 12, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18, 18, 18, 19, 19 };
        const int s = src[b];
        if (s == -1) return (par[0] + par[1]) * 0.5f;
        if (s == -2) return (par[2] + par[3]) * 0.5f;
        return par[s];
    }
    switch (b) {
    case 0:  return (2 * par[0] + par[1]) * 0.33333333f;
    case 1:  return (par[1] + 2 * par[2]) * 0.33333333f;
    case 2:  return (2 * par[3] + par[4]) * 0.33333333f;
    case 3:  return (par[4] + 2 * par[5]) * 0.33333333f;
    case 4:  return (par[6] + par[7]) * 0.5f;
    case 5:  return (par[8] + par[9]) * 0.5f;
    case 6:  return par[10];
    case 7:  return par[11];
    case 8:  return (par[12] + par[13]) * 0.5f;
    case 9:  return (par[14] + par[15]) * 0.5f;
    case 10: return par[16];
    case 11: return par[17];
    case 12: return par[18];
    case 13: return par[19];
    case 14: return (par[20] + par[21]) * 0.5f;
    case 15: return (par[22] + par[23]) * 0.5f;
    case 16: return (par[24] + par[25]) * 0.5f;
    case 17: return (par[26] + par[27]) * 0.5f;
    case 18: return (par[28] + par[29] + par[30] + par[31]) * 0.25f;
    case 19: return (par[32] + par[33]) * 0.5f;
    }
    return par[b];          // 34 -> 20 leaves par[20..33] untouched
}

// 13-tap complex FIR of hybrid6_cx / hybrid4_8_12_cx (aacps.c:310-321, :343-353).
// in: 13 consecutive complex slots, filt: [7][2]
__device__ __forceinline__ void hybrid_fir(const float *in, const float *filt, float &o_re, float &o_im)
{
    // (re, im) pairs: sum += f0 * (in0 + in1) + (-f1, f1) * swap(in0 - in1), the reference's terms
    // two per packed instruction
    const v2f *x = reinterpret_cast<const v2f *>(in);
    v2f sum = bc(filt[12]) * x[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const v2f in0 = x[j], in1 = x[12 - j];
        const v2f sm = in0 + in1, df = in0 - in1;
        const float f0 = filt[2 * j], f1 = filt[2 * j + 1];
        sum = sum + (bc(f0) * sm + v2f{-f1, f1} * __builtin_shufflevector(df, df, 1, 0));
    }
    o_re = sum.x;
    o_im = sum.y;
}

// One band, all 32 slots: decorrelation (aacps.c:696-753) fused with the mixing
// loop of stereo_processing (:900-969).
//   HEAVY = true : any band (all-pass chain, 14-slot or 1-slot delay)
//   HEAVY = false: bands >= 64 only, all of which use the 1-slot delay
//   input  : QMF bands -- this lane's column col[32] (register pairs);
//            sub-subbands (is_sub) -- LDS row w.sub[kh]
//   output : sub-subbands -> w.subL / w.subR rows; QMF column q -> X planes
// Envelope borders are walked once in ascending order (border[0] = -1,
// border[num_env] = 31, monotonic: what ff_ps_read_data produces).
// ALIGNED8: every border is 8k - 1 (frame_class 0, aacps.c:203-205), so the H
// interpolation can only restart at slots 0, 8, 16, 24 and the unrolled slot code in
// between is one straight-line block.
// DUAL (baseline layout, with HEAVY): the lanes whose first role is a sub-subband (their X
// store is discarded anyway) also carry the QMF band kh2 of their own column -- one of the
// bands >= 64 with the one-slot delay -- so that every X row leaves as one full 256-byte
// store and no separate pass over the slots is needed.  dual: this lane has a second role.
// hook(n): called at the head of every slot n (n is a compile-time constant at each call).  The
// fused kernel touches the next frame's records into L2 from it: early enough to be back before
// they are needed, late enough to survive in L2 until then.
typedef GBufT<2> GBufSO;          // the state record out (aux 2 = non-temporal): written once per frame, read by the next launch
typedef GBufT<2> GBufXR;          // the X rows of the slot loop: non-temporal too since they leave as whole (re, im) rows (-0.8 %; with the
                                  // plane layout of rounds 1-2, whose lines the hybrid synthesis completed later, it was +1.5 %)
template <bool HEAVY, bool ALIGNED8, bool DUAL, int X_BANDS = 64, class W, class Hook = NoHook>
__device__ __forceinline__ void ps_band(W &w, const float *__restrict__ g_tab, const signed char *kti,
                                        int is34, int kh, bool clear_state,
                                        const GBuf &SI, const GBufSO &SO, const GBufXR &X,
                                        bool is_sub, int q, const v2f (&col)[32],
                                        bool dual = false, int kh2 = 0, bool clear2 = false, Hook hook = Hook())
{
    static_assert(!DUAL || HEAVY, "the second role rides on the heavy pass");
    constexpr int dl_stride = 91 * 2, ap_stride = 50 * 2;
    constexpr int XC = HE_X_CHANNEL;                  // X record: [L, R][38][64][re, im]
    const int nr_allpass = is34 ? 50 : 30, short_delay = is34 ? 62 : 42;
    const int b = kti[kh];
    const int enable_ipdopd = W::IS_GENERAL ? w.p.enable_ipdopd : 0;
    const bool allpass = HEAVY && kh < nr_allpass;
    const bool d14 = HEAVY && !allpass && kh < short_delay;
    const v2f zero = {0.0f, 0.0f};

    // Complex values live in (re, im) register pairs; the arithmetic below is the
    // reference's, two products or sums per packed instruction.
    // Delay line: s[k][n - D] with D = 2 (all-pass input), 14 or 1: the lane's own column
    // for n >= D, before that the state tail hst[j] = s[k][j - 14].
    // Only the 14-slot bands read all of their tail; an all-pass band reads its last two slots, a one-slot band its
    // last.  The other lanes aim the load at their slot 13 again (one line instead of fourteen: 6 KB less per frame).
    v2f hst[14];
    const int first_slot = allpass ? 12 : d14 ? 0 : 13;
    // SLIM: slots 12 and 13 now (the all-pass and one-slot bands' whole tail), slots 0..11 -- which only the twelve
    // 14-slot bands read -- HST_AHEAD slots before their use inside the loop: four live pairs instead of fourteen
    constexpr int HST_AHEAD = 4;
    auto hst_load = [&](int j) {
        const int kv = opaque(kh * 8);
        const int js = (HEAVY && j < 13 && j < first_slot) ? 13 : j;
        const v2f t = (HEAVY && j < 13) ? SI.ldb2(kv + js * (dl_stride * 4), HEAAC_PS_DELAY)
                                        : SI.ldb2(kv, HEAAC_PS_DELAY + j * dl_stride);
        return clear_state ? zero : t;
    };
#pragma unroll
    for (int j = HEAVY ? 0 : 13; j < 14; j++) {
        if (W::SLIM && HEAVY && j >= HST_AHEAD && j < 12) continue;
        hst[j] = hst_load(j);
    }
    // all-pass history: ring of 5 per link, position = time mod 5 (state j = time j - 5)
    v2f ring[3][5];
    float ag[3] = {0, 0, 0};
    v2f qf[3] = {zero, zero, zero}, qfi[3] = {zero, zero, zero}, ph = zero, phi = zero;
#pragma unroll
    for (int m = 0; m < 3; m++)
#pragma unroll
        for (int j = 0; j < 5; j++) ring[m][j] = zero;
    if (allpass) {
        float g_decay_slope = 1.f - 0.05f * (kh - (is34 ? 32 : 10));
        g_decay_slope = g_decay_slope < 0.f ? 0.f : (g_decay_slope > 1.f ? 1.f : g_decay_slope);   // av_clipf
        const float a[3] = { 0.65143905753106f, 0.56471812200776f, 0.48954165955695f };
#pragma unroll
        for (int m = 0; m < 3; m++) {
            ag[m] = a[m] * g_decay_slope;
            const float qre = g_tab[TB_QFRACT + ((is34 * 50 + kh) * 3 + m) * 2];
            const float qim = g_tab[TB_QFRACT + ((is34 * 50 + kh) * 3 + m) * 2 + 1];
            qf[m] = v2f{qre, qim};
            qfi[m] = v2f{-qim, qre};                 // i * Q_fract
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const int kv = opaque(kh * 8);
                const v2f t = SI.ldb2(kv, HEAAC_PS_APDELAY + (m * 5 + j) * ap_stride);
                ring[m][j] = clear_state ? zero : t;
            }
        }
        const float phre = g_tab[TB_PHIFRACT + (is34 * 50 + kh) * 2];
        const float phim = g_tab[TB_PHIFRACT + (is34 * 50 + kh) * 2 + 1];
        ph = v2f{phre, phim};
        phi = v2f{-phim, phre};                      // i * phi_fract
    }
    // second role: delay tail, parameter band and H of band kh2
    const int b2 = DUAL ? kti[dual ? kh2 : 64] : 0;
    v2f hst2 = zero;
    if (DUAL) {
        const v2f t = SI.ldb2(opaque((dual ? kh2 : 64) * 8), HEAAC_PS_DELAY + 13 * dl_stride);
        hst2 = clear2 ? zero : t;
    }
    const float *tgrow2 = w.pw[b2];
    v2f hA2 = zero, hB2 = zero, hA2_step = zero, hB2_step = zero;
    const bool neg_im = (is34 && kh <= 13 && kh >= 9) || (!is34 && kh <= 1);
    const float *tgrow = w.pw[b];
    const v2f *srow = reinterpret_cast<const v2f *>(w.sub[is_sub ? kh : 0]);
    float *lrow = w.subL[is_sub ? kh : W::NSUB], *rrow = w.subR[is_sub ? kh : W::NSUB];   // row NSUB = scratch
    // column 0 of every row is rewritten by the hybrid synthesis at the end of the frame
    const int qs8 = q * 8;

    // H11/H12 and H21/H22 share a pair each, so one packed add steps two of them
    v2f hA = zero, hB = zero, hA_step = zero, hB_step = zero;        // (h11r, h12r), (h21r, h22r)
    v2f hAi = zero, hBi = zero, hAi_step = zero, hBi_step = zero;    // imaginary parts (IPD/OPD)
    int e = -1, stop = -1;
    v2f sub_back[2] = { zero, zero };

    // Fully unrolled over the 32 slots: ring positions and column indices are static.
#pragma unroll
    for (int n = 0; n < 32; n++) {
        hook(n);
        if constexpr (W::SLIM && HEAVY) {
            if (n + HST_AHEAD < 12) hst[n + HST_AHEAD] = hst_load(n + HST_AHEAD);
        }
        if ((!ALIGNED8 || (n & 7) == 0) && n > stop) {
            // next envelope (aacps.c:900-938)
            e++;
            const int start = __builtin_amdgcn_readfirstlane(w.p.border_position[e]);
            stop = __builtin_amdgcn_readfirstlane(w.p.border_position[e + 1]);
            const float width = 1.f / (stop - start);
            constexpr int R = W::IS_GENERAL ? 2 : 1;     // row step between H11, H12, H21, H22
            hA = v2f{w.Hs[e][0][b], w.Hs[e][R][b]};
            hB = v2f{w.Hs[e][2 * R][b], w.Hs[e][3 * R][b]};
            hA_step = (v2f{w.Hs[e + 1][0][b], w.Hs[e + 1][R][b]} - hA) * bc(width);
            hB_step = (v2f{w.Hs[e + 1][2 * R][b], w.Hs[e + 1][3 * R][b]} - hB) * bc(width);
            if constexpr (DUAL) {
                hA2 = v2f{w.Hs[e][0][b2], w.Hs[e][R][b2]};
                hB2 = v2f{w.Hs[e][2 * R][b2], w.Hs[e][3 * R][b2]};
                hA2_step = (v2f{w.Hs[e + 1][0][b2], w.Hs[e + 1][R][b2]} - hA2) * bc(width);
                hB2_step = (v2f{w.Hs[e + 1][2 * R][b2], w.Hs[e + 1][3 * R][b2]} - hB2) * bc(width);
            }
            if constexpr (W::IS_GENERAL) if (enable_ipdopd) {
                hAi = v2f{w.Hs[e][1][b], w.Hs[e][3][b]};
                hBi = v2f{w.Hs[e][5][b], w.Hs[e][7][b]};
                if (neg_im) { hAi = -hAi; hBi = -hBi; }
                hAi_step = (v2f{w.Hs[e + 1][1][b], w.Hs[e + 1][3][b]} - hAi) * bc(width);
                hBi_step = (v2f{w.Hs[e + 1][5][b], w.Hs[e + 1][7][b]} - hBi) * bc(width);
            }
        }
        // current sample and delayed sample s[k][n - D]
        v2f sv, dv;
        if (HEAVY) {
            // sub-subband lanes read their row (the others row 0, unused).  The reads are kept
            // unconditional: as select operands the compiler would put each one under an
            // exec-mask branch of its own.
            v2f subv = srow[n];
            v2f sub2;                                             // sub-subbands are all-pass bands: D = 2
            if constexpr (W::INPLACE_L) {
                // (the row is being overwritten by the left mix: the two samples back are kept in registers)
                sub2 = sub_back[n & 1];
                sub_back[n & 1] = subv;
                asm volatile("" : "+v"(subv));
            } else {
                sub2 = srow[n >= 2 ? n - 2 : 0];
                asm volatile("" : "+v"(subv), "+v"(sub2));
            }
            sv = is_sub ? subv : col[n];
            // state tail for the first slots (static register index per category)
            const v2f ap_d = n >= 2 ? (is_sub ? sub2 : col[n >= 2 ? n - 2 : 0]) : hst[12 + (n < 2 ? n : 0)];
            const v2f d14_d = n >= 14 ? col[n >= 14 ? n - 14 : 0] : hst[n < 14 ? n : 0];
            const v2f d1_d = n >= 1 ? col[n >= 1 ? n - 1 : 0] : hst[13];
            dv = allpass ? ap_d : d14 ? d14_d : d1_d;
        } else {
            sv = col[n];
            dv = n >= 1 ? col[n >= 1 ? n - 1 : 0] : hst[13];
        }
        const float tg = tgrow[n];
        v2f rv;
        if (HEAVY) {
            // all-pass chain, computed by every lane (no branch inside the slot);
            // lanes that are plain delays keep the delayed sample instead
            v2f x = bc(dv.x) * ph + bc(dv.y) * phi;            // (d.re ph.re - d.im ph.im, d.re ph.im + d.im ph.re)
#pragma unroll
            for (int m = 0; m < 3; m++) {
                const v2f a = bc(ag[m]) * x;
                // link_delay = 3, 4, 5: value written at time n - delay
                const int rp = (n + 5 - (3 + m)) % 5, wp = n % 5;
                const v2f ld = ring[m][rp];
                v2f nx = x;
                x = (bc(ld.x) * qf[m] + bc(ld.y) * qfi[m]) - a;
                nx += bc(ag[m]) * x;
                ring[m][wp] = nx;
            }
            rv = bc(tg) * (allpass ? x : dv);
        } else {
            rv = bc(tg) * dv;
        }

        hA += hA_step; hB += hB_step;
        // l = h11 s + h21 r,  r = h12 s + h22 r   (complex h only with IPD/OPD)
        v2f lv = bc(hA.x) * sv + bc(hB.x) * rv;
        v2f rr = bc(hA.y) * sv + bc(hB.y) * rv;
        if (enable_ipdopd) {
            hAi += hAi_step; hBi += hBi_step;
            const v2f si = rot90(sv), ri = rot90(rv);            // (-im, re)
            lv = (lv + bc(hAi.x) * si) + bc(hBi.x) * ri;
            rr = (rr + bc(hAi.y) * si) + bc(hBi.y) * ri;
        }
        // Branch-free stores: sub-subband lanes keep L/R in LDS rows (hybrid synthesis sums
        // them later) and send their global store to column 0, which the hybrid synthesis
        // rewrites afterwards; QMF lanes store to X and send their LDS store to a scratch row.
        if (HEAVY) {
            *reinterpret_cast<v2f *>(lrow + 2 * n) = lv;
            *reinterpret_cast<v2f *>(rrow + 2 * n) = rr;
            if constexpr (W::INPLACE_L) {
                // new delay-line tail = s[k][18..31], stored as the slots pass (the sub-subband rows do not survive the loop)
                if (n >= 18) SO.stb2(sv, opaque(kh * 8), HEAAC_PS_DELAY + (n - 18) * dl_stride);
            }
        }
        if constexpr (DUAL) {
            // second role (aacps.c:738-752 with the one-slot delay, then :940-969)
            const v2f d2 = n >= 1 ? col[n >= 1 ? n - 1 : 0] : hst2;
            const v2f rv2 = bc(tgrow2[n]) * d2;
            hA2 += hA2_step; hB2 += hB2_step;
            const v2f lv2 = bc(hA2.x) * col[n] + bc(hB2.x) * rv2;
            const v2f rr2 = bc(hA2.y) * col[n] + bc(hB2.y) * rv2;
            lv = dual ? lv2 : lv;
            rr = dual ? rr2 : rr;
        }
        // Bands from X_BANDS on are +0 in every slot and are not stored (ps_frame).  X_BANDS is a template parameter: as a
        // run-time value the lane mask lives in a scalar register pair across the whole unrolled loop and is spilled and
        // restored around every slot (+4 % kernel time); aiming the lanes past the buffer's range instead (the bounds
        // check drops the store) saves a quarter of what masking them saves.
        if (X_BANDS == 64 || q < X_BANDS) {
            const int qb = opaque(qs8);
            X.stb2(lv, qb, n * 128);
            X.stb2(rr, qb, XC + n * 128);
        }
        // bound the scheduler's look-ahead: without it the 32 unrolled slots are
        // interleaved until the register file overflows
        if ((n & (PS_SCHED_GROUP - 1)) == PS_SCHED_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
    }
    // new delay-line tail = s[k][18..31]
    if constexpr (!(HEAVY && W::INPLACE_L))
#pragma unroll
    for (int j = 0; j < 14; j++) {
        v2f v = col[18 + j];
        if (HEAVY) {
            const v2f t = srow[18 + j];
            v = is_sub ? t : v;
        }
        const int kv = opaque(kh * 8);
        SO.stb2(v, kv, HEAAC_PS_DELAY + j * dl_stride);
    }
    if (DUAL && dual) {
#pragma unroll
        for (int j = 0; j < 14; j++) SO.stb2(col[18 + j], opaque(kh2 * 8), HEAAC_PS_DELAY + j * dl_stride);
    }
    if (allpass) {
        // times 27..31 sit at ring positions (27 + j) % 5
#pragma unroll
        for (int m = 0; m < 3; m++)
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const int kv = opaque(kh * 8);
                SO.stb2(ring[m][(27 + j) % 5], kv, HEAAC_PS_APDELAY + (m * 5 + j) * ap_stride);
            }
    }
}

// power[i][n] = sum over the members of parameter band i (ascending hybrid band) of |s|^2.
// The member lists are constexpr, so both loops unroll into straight-line LDS reads.
template <bool IS34, int I>
__device__ __forceinline__ void band_power_one(const float *row, float *pw_col)
{
    constexpr int J0 = IS34 ? kMem34.first[I] : kMem20.first[I];
    constexpr int J1 = IS34 ? kMem34.first[I + 1] : kMem20.first[I + 1];
    float acc = 0.0f;
#pragma unroll
    for (int j = J0; j < J1; j++) {
        acc += row[IS34 ? kMem34.order[j] : kMem20.order[j]];      // |s|^2 = re*re + im*im
    }
    pw_col[I * 33] = acc;          // pw[I][n]
}
template <bool IS34, int I0, int I1>
__device__ __forceinline__ void band_power_range(const float *row, float *pw_col)
{
    if constexpr (I0 < I1) {
        band_power_one<IS34, I0>(row, pw_col);
        band_power_range<IS34, I0 + 1, I1>(row, pw_col);
    }
}
template <bool IS34>
__device__ __forceinline__ void band_power4(const float *row, float *pw_col, int quarter)
{
    constexpr BandMembers M = IS34 ? kMem34 : kMem20;
    if (quarter == 0)      band_power_range<IS34, M.quarter[0], M.quarter[1]>(row, pw_col);
    else if (quarter == 1) band_power_range<IS34, M.quarter[1], M.quarter[2]>(row, pw_col);
    else if (quarter == 2) band_power_range<IS34, M.quarter[2], M.quarter[3]>(row, pw_col);
    else                   band_power_range<IS34, M.quarter[3], M.quarter[4]>(row, pw_col);
}
__device__ __forceinline__ void band_power8_20(const float *row, float *pw_col, int part)
{
    constexpr BandMembers M = kMem20;
    if (part == 0)      band_power_range<false, M.eighth[0], M.eighth[1]>(row, pw_col);
    else if (part == 1) band_power_range<false, M.eighth[1], M.eighth[2]>(row, pw_col);
    else if (part == 2) band_power_range<false, M.eighth[2], M.eighth[3]>(row, pw_col);
    else if (part == 3) band_power_range<false, M.eighth[3], M.eighth[4]>(row, pw_col);
    else if (part == 4) band_power_range<false, M.eighth[4], M.eighth[5]>(row, pw_col);
    else if (part == 5) band_power_range<false, M.eighth[5], M.eighth[6]>(row, pw_col);
    else if (part == 6) band_power_range<false, M.eighth[6], M.eighth[7]>(row, pw_col);
    else                band_power_range<false, M.eighth[7], M.eighth[8]>(row, pw_col);
}
template <bool IS34>
__device__ __forceinline__ void band_power(const float *row, float *pw_col, int half)
{
    constexpr int SPLIT = IS34 ? kMem34.split : kMem20.split;
    constexpr int NPAR_ = IS34 ? 34 : 20;
    if (half == 0) band_power_range<IS34, 0, SPLIT>(row, pw_col);
    else           band_power_range<IS34, SPLIT, NPAR_>(row, pw_col);
}

// Which kernel variant owns a frame: the small-LDS one takes frames that are and
// were 20-band (and frames with PS off), the general one everything else.
__device__ __forceinline__ bool ps_frame_is_general(const HeaacPsFrame *g_p)
{
    // PS off (start == 0: plain L -> R copy) also goes to the general variant
    return !g_p->start || g_p->is34bands || g_p->is34bands_old || g_p->enable_ipdopd;
}

// FUSED: the mono QMF signal arrives in registers from the HF stage of the same wave
// (hfcol[n] = X[.][n][k] of band k = lane; the look-ahead slots 32..37 of the hybrid bands
// are already in w.inb) instead of being read from Xrec.
// FUSED also means: the caller has already copied the frame's PS record into w.p and the hybrid
// filters' history (in_buf state) into w.inb[.][0..5] (k_hfps issues those loads at the start of the
// frame, beside the HF stage's own).
template <bool GENERAL, bool FUSED = false, bool SLIM = false, class Hook = NoHook>
__device__ __forceinline__ void ps_frame(PsWaveT<GENERAL, SLIM> &w, const float *__restrict__ g_tab,
                                         const HeaacPsFrame *g_p, int top_qmf,
                                         const float *st_in, float *st_out,
                                         float *Xrec /* [2][38][64][re, im]: in: mono in [0], out: left, right */,
                                         int lane_in, int wave, const v2f (&hfcol)[32], Hook hook = Hook(),
                                         unsigned char *x_bands_out = nullptr, bool x_zero_above_top = false)
{
    static_assert(!(GENERAL && FUSED), "the fused path is the baseline layout only");
    static_assert(!SLIM || (FUSED && !GENERAL), "the slim layout is the fused kernel's");
    // `lane` is redefined opaquely at every phase: values derived from it (LDS addresses,
    // band indices) then live for one phase instead of being hoisted out of the frame loop
    // into registers that end up spilled.
    int lane = opaque(lane_in);
    using WT = PsWaveT<GENERAL, SLIM>;
    constexpr int XC = HE_X_CHANNEL;
    const GBuf SI(st_in), X(Xrec);
    const GBufXR XR(Xrec);
    const GBufSO SO(st_out);
    if constexpr (!FUSED) {
        const uint32_t *s = reinterpret_cast<const uint32_t *>(g_p);
        uint32_t *d = reinterpret_cast<uint32_t *>(&w.p);   // w.p is a reference into LDS
        for (int i = lane; i < (int)(sizeof(HeaacPsFrame) / 4); i += WAVE) d[i] = s[i];
    }
    wave_sync();
    STAMP(0);
    const HeaacPsFrame &p = w.p;

    if (!p.start) {
        // memcpy(sbr->X[1], sbr->X[0]) (aacsbr.c:1755); PS state untouched
        for (int t = lane; t < XC; t += WAVE) X.st(X.ld(t), t, XC);
        if (st_out != st_in)
            for (int t = lane; t < HEAAC_ST_PS; t += WAVE) SO.st(SI.ld(t), t);
        wave_sync();
        return;
    }

    const int is34 = GENERAL ? p.is34bands : 0;
    const int nr_bands = is34 ? 91 : 71, nr_par = is34 ? 34 : 20, nr_allpass = is34 ? 50 : 30;
    const int nsub = is34 ? 32 : 10, nlow = is34 ? 5 : 3;     // sub-subbands / hybrid QMF bands
    const int top = top_qmf + nr_bands - 64;                  // aacps.c:980
    const bool switched = GENERAL && is34 != p.is34bands_old;
    const BandMembers &M = is34 ? kMem34 : kMem20;

    // ---- every lane holds ONE QMF column (32 slots) in registers ----
    // general layout (20 or 34 bands):
    //   lanes [0, P2)    : q = 64 - P2 + lane        the bands of pass 2 (hybrid index 64 + lane)
    //   lanes [P2, nsub) : q = lane - P2             the nlow bands that feed the hybrid filters
    //   lanes [nsub, 64) : q = lane - nsub + nlow    hybrid index kh = lane (pass 1)
    // baseline layout (20 bands, the HF stage's own: lane = QMF band):
    //   lanes 0..2       : feed the hybrid filters;  pass 1: sub-subbands 0..2
    //   lanes 3..56      : pass 1: hybrid index lane + 7
    //   lanes 57..63     : pass 1: sub-subbands 3..9;  pass 2: hybrid index lane + 7 (their own column)
    const int P2 = nr_bands - 64;
    const int q_own = GENERAL ? (lane < P2 ? 64 - P2 + lane : lane < nsub ? lane - P2 : lane - nsub + nlow) : lane;
    v2f col[32];
    if constexpr (FUSED) {
#pragma unroll
        for (int n = 0; n < 32; n++) col[n] = hfcol[n];
    } else {
#pragma unroll
        for (int n = 0; n < 32; n++) {
            const int qb = opaque(q_own * 8);
            col[n] = X.ldb2(qb, n * 128);
        }
    }
    {
        // hybrid index of the lane's column (unless it is one of the nlow hybrid-filter inputs)
        const int kh_own = GENERAL ? (lane >= nsub ? lane : 64 + lane) : lane + 7;
        if (GENERAL ? (lane >= nsub || lane < P2) : lane >= 3) {
            if constexpr (!GENERAL && !SLIM) {
#pragma unroll
                for (int n = 0; n < 32; n++) w.pn[n * WT::PNS + kh_own] = col[n].x * col[n].x + col[n].y * col[n].y;
            }
        } else {
            // hybrid analysis input (aacps.c:362-367): in[i][j+6] = L[.][j][i]
#pragma unroll
            for (int n = 0; n < 32; n++) { w.inb[q_own][n + 6][0] = col[n].x; w.inb[q_own][n + 6][1] = col[n].y; }
        }
    }
    if constexpr (!FUSED)
    for (int t = lane; t < nlow * 6; t += WAVE) {
        const int i = t / 6, j = t % 6;
        w.inb[i][j][0] = SI.ld(t * 2, HEAAC_PS_INBUF);
        w.inb[i][j][1] = SI.ld(t * 2, HEAAC_PS_INBUF + 1);
        // lookahead slots 32..37
        w.inb[i][38 + j][0] = X.ld(((32 + j) * 64 + i) * 2);
        w.inb[i][38 + j][1] = X.ld(((32 + j) * 64 + i) * 2 + 1);
    }
    wave_sync();
    STAMP(1);
    lane = opaque(lane);
    // in_buf update (:391-394): in[i][0..5] <- in[i][32..37] = L[.][26..31][i], all 5 bands
    for (int t = lane; t < 5 * 6; t += WAVE) {
        const int i = t / 6, j = t % 6;
        float re = 0.0f, im = 0.0f;
        if (i < nlow) { re = w.inb[i][32 + j][0]; im = w.inb[i][32 + j][1]; }
        else if (!FUSED) { re = X.ld(((26 + j) * 64 + i) * 2); im = X.ld(((26 + j) * 64 + i) * 2 + 1); }
        if (i < nlow || !FUSED) {
            SO.st(re, t * 2, HEAAC_PS_INBUF);
            SO.st(im, t * 2, HEAAC_PS_INBUF + 1);
        }
    }
    if constexpr (FUSED) {
        // bands 3 and 4 (not hybrid bands in this layout): their lanes hold slots 26..31
        if (q_own == 3 || q_own == 4) {
#pragma unroll
            for (int j = 0; j < 6; j++) {
                SO.st(col[26 + j].x, (q_own * 6 + j) * 2, HEAAC_PS_INBUF);
                SO.st(col[26 + j].y, (q_own * 6 + j) * 2, HEAAC_PS_INBUF + 1);
            }
        }
    }
    // ---- parameter remapping (aacps.c:817-860), then every HBM / table read of the later
    // phases is issued here so that it is in flight during the hybrid filters, the band
    // powers and the transient detector: H-matrix LUT rows, H of the previous frame,
    // IPD/OPD histories, smoother state ----
    for (int t = lane; t < 5 * WT::NPAR; t += WAVE) {
        const int e = t / WT::NPAR, b = t % WT::NPAR;
        int iid = 0, icc = 0, ipd = 0, opd = 0;
        if (e < p.num_env && b < nr_par) {
            iid = remap_idx(p.iid_par[e], p.nr_iid_par, is34, b);
            icc = remap_idx(p.icc_par[e], p.nr_icc_par, is34, b);
            // remap34 / remap20 with full == 0 write 17 / 11 entries (aacps.c:461-499, 775-792).  With 17
            // phase parameters on the 20-band grid the reference's mixing loop goes on to b = 16 and reads
            // entries 11..16 of a local array it never wrote; they are 0 here (and in the oracle).
            if (p.enable_ipdopd && b < (is34 ? 17 : 11)) {
                ipd = remap_idx(p.ipd_par[e], p.nr_ipdopd_par, is34, b);
                opd = remap_idx(p.opd_par[e], p.nr_ipdopd_par, is34, b);
            }
        }
        w.iid_m[e][b] = (signed char)iid; w.icc_m[e][b] = (signed char)icc;
        if constexpr (GENERAL) { w.ipd_m[e][b] = (signed char)ipd; w.opd_m[e][b] = (signed char)opd; }
    }
    wave_sync();
    constexpr int NHL = (WT::NH * WT::NPAR + WAVE - 1) / WAVE;
    float hl[5][4], hrow[NHL], tr_peak = 0.0f, tr_smooth = 0.0f, tr_diff = 0.0f;
    int opd_hist0 = 0, ipd_hist0 = 0;
#pragma unroll
    for (int e = 0; e < 5; e++) hl[e][0] = hl[e][1] = hl[e][2] = hl[e][3] = 0.0f;
    if (lane < 34) {
        const signed char *hist = reinterpret_cast<const signed char *>(st_in + HEAAC_PS_HIST);
        opd_hist0 = hist[lane]; ipd_hist0 = hist[34 + lane];
        tr_peak = SI.ld(lane, HEAAC_PS_PEAK); tr_smooth = SI.ld(lane, HEAAC_PS_PSMOOTH);
        tr_diff = SI.ld(lane, HEAAC_PS_PDIFF);
    }
    auto fetch_lut = [&]() {
      if (lane < nr_par) {
        const int b = lane;
        const float *LUT = g_tab + ((p.icc_mode < 3) ? TB_HA : TB_HB);
#pragma unroll
        for (int e = 0; e < 5; e++) {
            const int ee = e < p.num_env ? e : 0;
            // (row and column clamped to the table: heaac_dsp.h, record validation)
            int row = w.iid_m[ee][b] + 7 + 23 * (p.iid_quant ? 1 : 0), colx = w.icc_m[ee][b];
            row = row < 0 ? 0 : row > 45 ? 45 : row;
            colx = colx < 0 ? 0 : colx > 7 ? 7 : colx;
            const float4 h4 = *reinterpret_cast<const float4 *>(LUT + (row * 8 + colx) * 4);
            hl[e][0] = h4.x; hl[e][1] = h4.y; hl[e][2] = h4.z; hl[e][3] = h4.w;
        }
      }
    };
    // (early, to land during the hybrid filters and the band powers -- unless registers are what is short)
    if constexpr (!SLIM) fetch_lut();
#pragma unroll
    for (int i = 0; i < NHL; i++) {
        const int t = lane + WAVE * i, j = t / WT::NPAR, b = t % WT::NPAR;
        hrow[i] = t < WT::NH * WT::NPAR ? SI.ld((GENERAL ? j : 2 * j) * 34 + b, HEAAC_PS_H) : 0.0f;
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- hybrid filters -> sub[ks][n] ----
    auto hybrid_put = [&](int ks, int n, float re, float im) {
        w.sub[ks][2 * n] = re;
        w.sub[ks][2 * n + 1] = im;
        if constexpr (!GENERAL && !SLIM) w.pn[n * WT::PNS + ks] = re * re + im * im;
    };
    if (!is34) {
        // 20-band layout: 10 sub-subbands x 32 slots = five passes of the wave; pass `it` forms the
        // sub-subbands 2 it and 2 it + 1 (one per half-wave), so which filter a pass applies is known
        // at compile time and the LDS reads of all passes are in flight together.
        const int h = lane >> 5, n = lane & 31;
        const float *F = w.hybF, *G = w.hybG;
#pragma unroll
        for (int it = 0; it < 5; it++) {
            const int ks = 2 * it + h;
            float re, im;
            if (it < 3) {
                // hybrid6_cx (:303-336): out = temp[fa] (+ temp[fb]); order 6,7,0,1,2+5,3+4
                const float *in = &w.inb[0][n][0];
                const int fa = it == 0 ? 6 + h : it == 1 ? h : 2 + h;
                hybrid_fir(in, F + fa * 14, re, im);
                if (it == 2) {
                    float br, bi;
                    hybrid_fir(in, F + (5 - h) * 14, br, bi);
                    re = re + br;
                    im = im + bi;
                }
            } else {
                // hybrid2_re (:283-301): band 1 reversed, band 2 not; half-wave h forms out[h]
                const int reverse = it == 3 ? 1 : 0;
                const float *in = &w.inb[it - 2][n][0];
                const float re_in = G[6] * in[12], im_in = G[6] * in[13];
                float re_op = 0.0f, im_op = 0.0f;
#pragma unroll
                for (int j = 0; j < 6; j += 2) {
                    re_op += G[j + 1] * (in[2 * (j + 1)] + in[2 * (12 - j - 1)]);
                    im_op += G[j + 1] * (in[2 * (j + 1) + 1] + in[2 * (12 - j - 1) + 1]);
                }
                // out[reverse] = in + op, out[!reverse] = in - op: a - b is a + (-b) exactly
                const bool plus = h == reverse;
                re = re_in + (plus ? re_op : -re_op);
                im = im_in + (plus ? im_op : -im_op);
            }
            hybrid_put(ks, n, re, im);
        }
    } else if constexpr (GENERAL) {
        for (int t = lane; t < 32 * 32; t += WAVE) {
            const int ks = t >> 5, n = t & 31;
            float re, im;
            int qb, f, off;
            if (ks < 12)      { qb = 0; f = ks;      off = TB_F34_0_12; }
            else if (ks < 20) { qb = 1; f = ks - 12; off = TB_F34_1_8; }
            else              { qb = 2 + ((ks - 20) >> 2); f = (ks - 20) & 3; off = TB_F34_2_4; }
            hybrid_fir(&w.inb[qb][n][0], g_tab + off + f * 14, re, im);
            hybrid_put(ks, n, re, im);
        }
    }
    wave_sync();

    STAMP(2);
    lane = opaque(lane);
    // ---- band power (aacps.c:673-678): members of each parameter band in ascending
    // hybrid-band order.  The member lists are compile-time constants, so the sums
    // unroll into straight-line LDS reads; two half-waves split the parameter bands.
    if constexpr (GENERAL) {
        // sixteen slots at a time: |s|^2 of the lane's column and (lanes below nsub) of its sub-subband row, then the
        // sums with four lanes per slot, each a quarter of the parameter bands
        const int kh_own = lane >= nsub ? lane : 64 + lane;
        const bool has_col = lane >= nsub || lane < P2;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            if (has_col) {
#pragma unroll
                for (int n = 0; n < 16; n++) {
                    const v2f c = col[16 * r + n];
                    w.pn[n * WT::PNS + kh_own] = c.x * c.x + c.y * c.y;
                }
            }
            if (lane < nsub) {
#pragma unroll
                for (int n = 0; n < 16; n++) {
                    const float re = w.sub[lane][2 * (16 * r + n)], im = w.sub[lane][2 * (16 * r + n) + 1];
                    w.pn[n * WT::PNS + lane] = re * re + im * im;
                }
            }
            wave_sync();
            const int n = lane & 15, quarter = lane >> 4;
            const float *row = w.pn + n * WT::PNS;
            if (is34) band_power4<true>(row, &w.pw[0][16 * r + n], quarter);
            else      band_power4<false>(row, &w.pw[0][16 * r + n], quarter);
            wave_sync();
        }
    } else if constexpr (SLIM) {
        // eight slots at a time: |s|^2 of the lane's column (lanes 3..63: hybrid bands 10..70) and of the ten
        // sub-subband rows, then the sums with eight lanes per slot, each an eighth of the parameter bands' members
        const int kh_own = lane + 7;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (lane >= 3) {
#pragma unroll
                for (int n = 0; n < 8; n++) {
                    const v2f c = col[8 * r + n];
                    w.pn[n * WT::PNS + kh_own] = c.x * c.x + c.y * c.y;
                }
            }
            if (lane < 10) {
#pragma unroll
                for (int n = 0; n < 8; n++) {
                    const float re = w.sub[lane][2 * (8 * r + n)], im = w.sub[lane][2 * (8 * r + n) + 1];
                    w.pn[n * WT::PNS + lane] = re * re + im * im;
                }
            }
            wave_sync();
            const int n = lane & 7, part = lane >> 3;
            band_power8_20(w.pn + n * WT::PNS, &w.pw[0][8 * r + n], part);
            wave_sync();
        }
    } else {
        const int n = lane & 31, half = lane >> 5;
        const float *row = w.pn + n * WT::PNS;
        band_power<false>(row, &w.pw[0][n], half);
        wave_sync();
    }
    STAMP(3);
    lane = opaque(lane);
    // ---- transient detection (:681-692), one lane per parameter band ----
    if (lane < nr_par) {
        const int i = lane;
        float peak = tr_peak, smooth = tr_smooth, diff = tr_diff;
        if (switched) { peak = 0.0f; smooth = 0.0f; diff = 0.0f; }
        // (the slim layout reads the powers eight slots at a time: the QMF column holds 64 registers meanwhile)
        constexpr int PG = SLIM ? 8 : 32;
        float prow[PG];
#pragma unroll
        for (int n = 0; n < 32; n++) {
            if (n % PG == 0) {
#pragma unroll
                for (int q = 0; q < PG; q++) prow[q] = w.pw[i][n + q];
                if (SLIM) __builtin_amdgcn_sched_barrier(0);
            }
            const float pwr = prow[n % PG];
            const float decayed_peak = 0.76592833836465f * peak;
            peak = decayed_peak > pwr ? decayed_peak : pwr;
            smooth += 0.25f * (pwr - smooth);
            diff += 0.25f * (peak - pwr - diff);
            const float denom = 1.5f * diff;
            w.pw[i][n] = (denom > smooth) ? smooth / denom : 1.0f;
        }
        SO.st(peak, i, HEAAC_PS_PEAK);
        SO.st(smooth, i, HEAAC_PS_PSMOOTH);
        SO.st(diff, i, HEAAC_PS_PDIFF);
    } else if (lane < 34) {
        // parameter bands 20..33 are not touched in 20-band mode
        const int i = lane;
        SO.st(switched ? 0.0f : tr_peak, i, HEAAC_PS_PEAK);
        SO.st(switched ? 0.0f : tr_smooth, i, HEAAC_PS_PSMOOTH);
        SO.st(switched ? 0.0f : tr_diff, i, HEAAC_PS_PDIFF);
    }

    STAMP(4);
    lane = opaque(lane);
    // ---- parameter remapping + H matrices (aacps.c:817-899) ----
    // row 0 = H of the last envelope of the previous frame, remapped on a 20<->34 switch
#pragma unroll
    for (int i = 0; i < NHL; i++) {
        const int t = lane + WAVE * i, j = t / WT::NPAR, b = t % WT::NPAR;
        if (t < WT::NH * WT::NPAR) {
            float v = hrow[i];
            if (switched) v = remap_val(st_in + HEAAC_PS_H + (GENERAL ? j : 2 * j) * 34, is34, b);
            w.Hs[0][j][b] = v;                 // baseline keeps the real rows only
        }
    }
    wave_sync();
    if constexpr (SLIM) fetch_lut();
    bool h_ok = true;
    float h_prev[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
    if (lane < nr_par) {
        const int b = lane;
        int opd_hist = opd_hist0, ipd_hist = ipd_hist0;
        if (switched && b < 17) { opd_hist = 0; ipd_hist = 0; }        // ipdopd_reset
        if constexpr (FUSED) {
            // row 0 = the previous frame's matrices: the first end of the first envelope's interpolation
            h_prev[0] = w.Hs[0][0][b]; h_prev[1] = w.Hs[0][1][b]; h_prev[2] = w.Hs[0][2][b]; h_prev[3] = w.Hs[0][3][b];
        }
        // (the LUT rows hl[e] of every envelope were fetched ahead; the IPD/OPD history
        // chain runs over them)
#pragma unroll
        for (int e = 0; e < 5; e++) {
            if (e >= p.num_env) break;
            float h11 = hl[e][0], h12 = hl[e][1], h21 = hl[e][2], h22 = hl[e][3];
            float h11i = 0.0f, h12i = 0.0f, h21i = 0.0f, h22i = 0.0f;
            if (GENERAL && p.enable_ipdopd && b < p.nr_ipdopd_par) {
                const int opd_idx = (opd_hist * 8 + w.opd_m[e][b]) & 511;      // (& 511: no-op on valid indices)
                const int ipd_idx = (ipd_hist * 8 + w.ipd_m[e][b]) & 511;
                const float opd_re = g_tab[TB_PD_RE + opd_idx], opd_im = g_tab[TB_PD_IM + opd_idx];
                const float ipd_re = g_tab[TB_PD_RE + ipd_idx], ipd_im = g_tab[TB_PD_IM + ipd_idx];
                opd_hist = opd_idx & 0x3F;
                ipd_hist = ipd_idx & 0x3F;
                const float ipd_adj_re = opd_re * ipd_re + opd_im * ipd_im;
                const float ipd_adj_im = opd_im * ipd_re - opd_re * ipd_im;
                h11i = h11 * opd_im;     h11 = h11 * opd_re;
                h12i = h12 * ipd_adj_im; h12 = h12 * ipd_adj_re;
                h21i = h21 * opd_im;     h21 = h21 * opd_re;
                h22i = h22 * ipd_adj_im; h22 = h22 * ipd_adj_re;
            }
            if constexpr (FUSED) {
                // (see x_bands below) one factor of h11 / h21 and one of h12 / h22 safely positive over the envelope
                const float hn[4] = { h11, h12, h21, h22 };
                bool bounded = true;
#pragma unroll
                for (int j = 0; j < 4; j++) bounded = bounded && fabsf(hn[j]) <= 16.0f && fabsf(h_prev[j]) <= 16.0f;
                const bool left = (h_prev[0] > 1e-3f && hn[0] > 1e-3f) || (h_prev[2] > 1e-3f && hn[2] > 1e-3f);
                const bool right = (h_prev[1] > 1e-3f && hn[1] > 1e-3f) || (h_prev[3] > 1e-3f && hn[3] > 1e-3f);
                h_ok = h_ok && bounded && left && right;
#pragma unroll
                for (int j = 0; j < 4; j++) h_prev[j] = hn[j];
            }
            if constexpr (GENERAL) {
                w.Hs[e + 1][0][b] = h11; w.Hs[e + 1][1][b] = h11i;
                w.Hs[e + 1][2][b] = h12; w.Hs[e + 1][3][b] = h12i;
                w.Hs[e + 1][4][b] = h21; w.Hs[e + 1][5][b] = h21i;
                w.Hs[e + 1][6][b] = h22; w.Hs[e + 1][7][b] = h22i;
            } else {
                w.Hs[e + 1][0][b] = h11; w.Hs[e + 1][1][b] = h12;
                w.Hs[e + 1][2][b] = h21; w.Hs[e + 1][3][b] = h22;
            }
        }
        // new history (bytes of two packed rows)
        signed char *ho = reinterpret_cast<signed char *>(st_out + HEAAC_PS_HIST);
        ho[b] = (signed char)opd_hist;
        ho[34 + b] = (signed char)ipd_hist;
    } else if (lane < 34) {
        signed char *ho = reinterpret_cast<signed char *>(st_out + HEAAC_PS_HIST);
        ho[lane] = (signed char)opd_hist0;
        ho[34 + lane] = (signed char)ipd_hist0;
    }
    if (lane == 0) {
        signed char *ho = reinterpret_cast<signed char *>(st_out + HEAAC_PS_HIST);
        ho[68] = ho[69] = ho[70] = ho[71] = 0;
    }
    wave_sync();
    // H state out: real rows always, imaginary rows only while IPD/OPD is on
    if constexpr (GENERAL) {
        for (int t = lane; t < 8 * 34; t += WAVE) {
            const int j = t / 34, b = t % 34;
            const bool imag = j & 1;
            float v;
            if (!imag || p.enable_ipdopd) v = b < nr_par ? w.Hs[p.num_env][j][b] : 0.0f;
            else v = SI.ld(t, HEAAC_PS_H);
            SO.st(v, t, HEAAC_PS_H);
        }
    } else {
        for (int t = lane; t < 4 * 34; t += WAVE) {
            const int j = t / 34, b = t % 34;
            SO.st(b < 20 ? w.Hs[p.num_env][j][b] : 0.0f, 2 * j * 34 + b, HEAAC_PS_H);
        }
        if (st_out != st_in)
            for (int t = lane; t < 4 * 34; t += WAVE) {
                const int j = t / 34, b = t % 34;
                SO.st(SI.ld((2 * j + 1) * 34 + b, HEAAC_PS_H), (2 * j + 1) * 34 + b, HEAAC_PS_H);
            }
    }

    STAMP(5);
    lane = opaque(lane);
    // every border at 8k - 1 (what frame_class 0 produces): fast straight-line variant
    bool aligned8 = true;
    for (int e = 1; e <= p.num_env; e++) aligned8 = aligned8 && ((p.border_position[e] & 7) == 7);
    aligned8 = __builtin_amdgcn_readfirstlane(aligned8);
    // ---- which X bands leave the wave (fused kernel, baseline layout) ----
    // Above `top` the mono signal is +0 (sbr_x_gen writes the literal there; the caller says whether the first slots,
    // which follow the PREVIOUS frame's range, agree) and so is its delay line (cleared, aacps.c:980-983); none of those
    // bands runs the all-pass chain once top >= 23.  What the mixing makes of them is h11 (+0) + h21 (+0): +0 unless
    // BOTH factors are negative (or not finite), and likewise h12 / h22 for the right channel.  If, for every
    // parameter band that covers such a band and every envelope, one factor of each pair stays above 1e-3 at both ends
    // of its interpolation (and all four inside +-16, so that 32 rounded steps cannot carry it below zero: h_ok, formed
    // with the matrices above), every X value of the bands from top (rounded up to a 128-byte line of a row: 16 bands) on is exactly
    // +0: they are not stored, the frame's byte in x_bands_out says so, and k_synth reads them from a page of zeros.
    int x_bands = 64;
    if constexpr (FUSED) {
        const int top16 = (top_qmf + 15) & ~15;
        // (only the straight-line slot loop has the store variants: every border at 8k - 1, what frame class 0 gives)
        if (x_zero_above_top && aligned8 && top16 < 64 && top16 >= 32 && p.num_env >= 1) {
            const bool bad = lane < nr_par && lane >= kMem20.kti[top16 + 7] && !h_ok;
            if (__builtin_amdgcn_readfirstlane(__ballot(bad) == 0ull)) x_bands = top16;
        }
        if (x_bands_out && x_bands != 64 && lane == 0) { x_bands_out[0] = (unsigned char)x_bands; x_bands_out[1] = (unsigned char)x_bands; }
    }
    // ---- pass 1: hybrid bands 0..63 = all sub-subbands + the first QMF bands ----
    {
        const int kh = GENERAL ? lane : (lane < 3 ? lane : lane >= 57 ? lane - 54 : lane + 7);
        const bool is_sub = kh < nsub;
        // sub-subband lanes send their (discarded) X store to their own column: pass 2 or the
        // hybrid synthesis rewrites it afterwards
        const int qcol = GENERAL ? (is_sub ? 0 : kh - nsub + nlow) : lane;
        if constexpr (GENERAL) {
            if (aligned8)
                ps_band<true, true, false>(w, g_tab, M.kti, is34, kh, switched || kh >= top, SI, SO, XR, is_sub, qcol, col);
            else
                ps_band<true, false, false>(w, g_tab, M.kti, is34, kh, switched || kh >= top, SI, SO, XR, is_sub, qcol, col);
        } else {
            // lanes 57..63 also carry the bands 64..70 of their own columns
            const bool dual = lane >= 57;
            const int kh2 = lane + 7;
            if (aligned8 && x_bands == 48)
                ps_band<true, true, true, 48>(w, g_tab, M.kti, is34, kh, kh >= top, SI, SO, XR, is_sub, qcol, col,
                                              dual, kh2, kh2 >= top, hook);
            else if (aligned8 && x_bands == 32)
                ps_band<true, true, true, 32>(w, g_tab, M.kti, is34, kh, kh >= top, SI, SO, XR, is_sub, qcol, col,
                                              dual, kh2, kh2 >= top, hook);
            else if (aligned8)
                ps_band<true, true, true>(w, g_tab, M.kti, is34, kh, kh >= top, SI, SO, XR, is_sub, qcol, col,
                                          dual, kh2, kh2 >= top, hook);
            else
                ps_band<true, false, true>(w, g_tab, M.kti, is34, kh, kh >= top, SI, SO, XR, is_sub, qcol, col,
                                           dual, kh2, kh2 >= top, hook);
        }
    }
    STAMP(6);
    lane = opaque(lane);
    // ---- pass 2: hybrid bands 64.. (all use the one-slot delay) ----
    if constexpr (GENERAL) if (lane < nr_bands - 64) {
        const int kh = 64 + lane;
        if (aligned8)
            ps_band<false, true, false>(w, g_tab, M.kti, is34, kh, switched || kh >= top, SI, SO, XR,
                                        false, kh - nsub + nlow, col);
        else
            ps_band<false, false, false>(w, g_tab, M.kti, is34, kh, switched || kh >= top, SI, SO, XR,
                                         false, kh - nsub + nlow, col);
    }
    STAMP(7);
    lane = opaque(lane);
    // bands that exist in the state record but not in this layout / all-pass set
    if (st_out != st_in || switched)
    for (int t = lane; t < 14 * 91; t += WAVE) {
        const int k = t % 91;
        if (k >= nr_bands) {
            const float a = SI.ld(t * 2, HEAAC_PS_DELAY), b = SI.ld(t * 2, HEAAC_PS_DELAY + 1);
            SO.st(switched ? 0.0f : a, t * 2, HEAAC_PS_DELAY);
            SO.st(switched ? 0.0f : b, t * 2, HEAAC_PS_DELAY + 1);
        }
    }
    if (st_out != st_in || switched)
    for (int t = lane; t < 15 * 50; t += WAVE) {
        const int k = t % 50;
        if (k >= nr_allpass) {
            const float a = SI.ld(t * 2, HEAAC_PS_APDELAY), b = SI.ld(t * 2, HEAAC_PS_APDELAY + 1);
            SO.st(switched ? 0.0f : a, t * 2, HEAAC_PS_APDELAY);
            SO.st(switched ? 0.0f : b, t * 2, HEAAC_PS_APDELAY + 1);
        }
    }
    wave_sync();

    STAMP(8);
    lane = opaque(lane);
    // ---- hybrid synthesis (aacps.c:397-445) for the lowest QMF bands ----
    {
        const int n = lane & 31, side = lane >> 5;
        const float *rows = side ? &w.subR[0][0] : &w.subL[0][0];
        const int o0 = side * XC + n * 128, o1 = o0 + 1;       // (re, im) of band qq at o0 + 2 qq, o1 + 2 qq
#define SUBV(i, c) rows[(i) * SUB_STRIDE + 2 * n + (c)]
        if (is34) {
            const int first[5] = { 0, 12, 20, 24, 28 }, cnt[5] = { 12, 8, 4, 4, 4 };
#pragma unroll
            for (int qq = 0; qq < 5; qq++) {
                float re = 0.0f, im = 0.0f;
                for (int i = 0; i < cnt[qq]; i++) { re += SUBV(first[qq] + i, 0); im += SUBV(first[qq] + i, 1); }
                X.st(re, o0, 2 * qq);
                X.st(im, o1, 2 * qq);
            }
        } else {
            X.st(SUBV(0, 0) + SUBV(1, 0) + SUBV(2, 0) + SUBV(3, 0) + SUBV(4, 0) + SUBV(5, 0), o0, 0);
            X.st(SUBV(0, 1) + SUBV(1, 1) + SUBV(2, 1) + SUBV(3, 1) + SUBV(4, 1) + SUBV(5, 1), o1, 0);
            X.st(SUBV(6, 0) + SUBV(7, 0), o0, 2);
            X.st(SUBV(6, 1) + SUBV(7, 1), o1, 2);
            X.st(SUBV(8, 0) + SUBV(9, 0), o0, 4);
            X.st(SUBV(8, 1) + SUBV(9, 1), o1, 4);
        }
#undef SUBV
    }
    wave_sync();
    STAMP(9);
}

