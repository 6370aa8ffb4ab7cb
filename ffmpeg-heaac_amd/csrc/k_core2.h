// k_core2.h -- AAC core synthesis, two channels per wavefront, FFT in registers.
//
// imdct_and_windowing() (aacdec.c:1741-1806) as in k_core.h, but the 512-point (or
// 8 x 64-point) split-radix FFT of ff_imdct_half (mdct.c:124-159, fft.c:283-351) no
// longer makes one LDS round trip per pass level.  Each HALF-wave (32 lanes) owns one
// channel and keeps its 512 complex points in registers, 16 per lane, in three layouts
// chosen so that every butterfly of a group of pass levels is lane-local:
//
//   layout A  lane h holds z[16h + i]            fft4, fft8, pass16     (fft.c:292-339)
//   layout B  lane (c = h / 4, r = h % 4) holds z[64c + r + 4m]   pass32, pass64
//   layout C  lane l holds z[l + 32j]            pass128, pass256, pass512
//
// i, m, j = 0..15 are register indices.  Two padded LDS transposes (A -> B, B -> C)
// replace the seven pass-level round trips; the butterflies, their operands and their
// order inside every output are the reference's, so results are bit-identical.
//
// Which 16- and 64-element chunks are complete FFT blocks (fft16 / fft64) and which are
// pairs of half-size blocks follows from the split-radix recursion; the masks are
// derived from the same SrSchedule the LDS version walks.
#pragma once
#include "k_common.h"
#include "k_core.h"
#include "heaac_dsp.h"

#define C2_TSTRIDE 544            // complex slots per channel region: 512 + padding

struct Core2Lds {
    float tab[CORE_TAB_FLOATS];   // blob range [TB_COS16, TB_SINE_SHORT + 128)
    float2 rotA512[512];          // [i][h]: (tcos[k], tsin[k]) of the k with revtab[k] = 16h + i
    float2 rotA64[64];            // [i][c]: same for the 64-point transform, chunk c of a window
    uint16_t kA512[512];
    uint16_t kA64[64];
};

__device__ __forceinline__ void core2_lds_init(Core2Lds &L, const float *g_tab, const uint16_t *g_rev)
{
    wg_copy_f4(L.tab, g_tab, CORE_TAB_FLOATS);
    for (int k = threadIdx.x; k < 512; k += blockDim.x) {
        const int e = g_rev[RV_512 + k], slot = (e & 15) * 32 + (e >> 4);
        L.kA512[slot] = (uint16_t)k;
        L.rotA512[slot] = make_float2(g_tab[TB_ROT2048 + k], g_tab[TB_ROT2048 + 512 + k]);
    }
    for (int k = threadIdx.x; k < 64; k += blockDim.x) {
        const int e = g_rev[RV_64 + k], slot = (e & 15) * 4 + (e >> 4);
        L.kA64[slot] = (uint16_t)k;
        L.rotA64[slot] = make_float2(g_tab[TB_ROT256 + k], g_tab[TB_ROT256 + 64 + k]);
    }
}

// The tables of the transform + windowing code as pointers: a kernel decides per table whether it reads its LDS copy or the
// blob in global memory (k_core_ana keeps the 16 KB of long windows and N = 2048 rotation tables out of LDS for a
// seventh wave per CU; every pointer that one expression selects between must be in ONE address space).
struct CoreTabs {
    const float *cos;                 // ff_cos_16 .. ff_cos_512 at their blob offsets
    const float *rot2048, *rot256;    // tcos[n/4] then tsin[n/4]
    const float *kbd_long, *sine_long, *kbd_short, *sine_short;
    const float2 *rotA512, *rotA64;   // pre-rotation twiddles in layout-A order
    const uint16_t *kA512, *kA64;     // the k behind every layout-A slot
};
__device__ __forceinline__ CoreTabs core2_tabs(const Core2Lds &L)
{
    return CoreTabs{ L.tab, L.tab + TB_ROT2048, L.tab + TB_ROT256, L.tab + TB_KBD_LONG, L.tab + TB_SINE_LONG,
                     L.tab + TB_KBD_SHORT, L.tab + TB_SINE_SHORT, L.rotA512, L.rotA64, L.kA512, L.kA64 };
}

// bit c of the result: a block of size 1 << bits starts at offset c << bits
constexpr unsigned sr_block_mask(const SrSchedule &s, int bits)
{
    unsigned m = 0;
    for (int i = 0; i < s.cnt[bits]; i++) m |= 1u << (s.off[bits][i] >> bits);
    return m;
}
constexpr unsigned kMask16_512 = sr_block_mask(kSched512, 4);
constexpr unsigned kMask64_512 = sr_block_mask(kSched512, 6);
constexpr unsigned kMask16_64  = sr_block_mask(kSched64, 4);
static_assert(sr_block_mask(kSched512, 7) == 0x0d, "pass128 blocks at 0, 256, 384");
static_assert(sr_block_mask(kSched512, 8) == 0x01 && sr_block_mask(kSched512, 9) == 0x01, "one 256 / 512 block");
static_assert(sr_block_mask(kSched64, 6) == 0x01 && kMask16_64 == 0x0d, "fft64 = fft32, fft16, fft16");

// TRANSFORM or TRANSFORM_ZERO chosen per lane (k == 0 is the reference's multiplication-free
// first butterfly, fft.c:264; selecting its operands keeps signed zeros and non-finite
// inputs exactly as the reference has them).
__device__ __forceinline__ void sr_transform_sel(cpx &a0, cpx &a1, cpx &a2, cpx &a3,
                                                 float wre, float wim, bool zero)
{
    float t1 = a2.re * wre + a2.im * wim;
    float t2 = a2.im * wre - a2.re * wim;
    float t5 = a3.re * wre - a3.im * wim;
    float t6 = a3.im * wre + a3.re * wim;
    t1 = zero ? a2.re : t1;  t2 = zero ? a2.im : t2;
    t5 = zero ? a3.re : t5;  t6 = zero ? a3.im : t6;
    float t3 = t5 - t1;  t5 = t5 + t1;
    a2.re = a0.re - t5;  a0.re = a0.re + t5;
    a3.im = a1.im - t3;  a1.im = a1.im + t3;
    float t4 = t2 - t6;  t6 = t2 + t6;
    a3.re = a1.re - t4;  a1.re = a1.re + t4;
    a2.im = a0.im - t6;  a0.im = a0.im + t6;
}

// ff_imdct_half for one channel per half-wave: N = 2048 (eight == false) or 8 x N = 256.
//   in  : the channel's 1024 coefficients in LDS (may be the same memory as T)
//   T   : the channel's transpose region, C2_TSTRIDE complex slots; on return it holds
//         buf[1024] (the reference's buf[], floats) in its first 4 KiB
//   hl  : lane within the half-wave (0..31)
// Both half-waves of a wave call this together (each with its own in / T / eight).
// Twiddles of the long transform that depend on the lane only -- the pre-rotation pairs of layout A and the post-rotation
// pairs (tsin[e], tcos[e]), e = hl + 32 j: a kernel whose tables sit in global memory loads them once and keeps them in
// registers across its units (LongTw, TW = true); TW = false reads the tables at every unit.
struct LongTw { float2 pre[16], post[16]; };
__device__ __forceinline__ void core2_load_long_twiddles(const CoreTabs &L, int hl, LongTw &t)
{
#pragma unroll
    for (int i = 0; i < 16; i++) {
        t.pre[i] = L.rotA512[hl + 32 * i];
        t.post[i] = make_float2(L.rot2048[512 + hl + 32 * i], L.rot2048[hl + 32 * i]);
    }
}

template <bool TW>
__device__ __forceinline__ void imdct_half_regs_tw(const CoreTabs &L, const float *in, cpx *T, bool eight, int hl,
                                                   const LongTw &tw)
{
    cpx z[16];
    const int c = hl >> 2, r = hl & 3;       // layout B: 64-chunk (= window when eight) and residue
    // ---- pre-rotation straight into layout A (mdct.c:136-141) ----
    {
        const uint16_t *kt = eight ? L.kA64 + (hl & 3) : L.kA512 + hl;
        const float2 *rt = eight ? L.rotA64 + (hl & 3) : L.rotA512 + hl;
        const int kstride = eight ? 4 : 32;
        const float *x = eight ? in + (hl >> 2) * 128 : in;
        const int last = eight ? 127 : 1023;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int k = kt[i * kstride];
            float2 w;
            if constexpr (TW) {
                // (the short transform's pairs are read only by the half-waves that need them)
                w = tw.pre[i];
                if (eight) w = rt[i * kstride];
            } else {
                w = rt[i * kstride];
            }
            cmul(z[i].re, z[i].im, x[last - 2 * k], x[2 * k], w.x, w.y);
        }
    }
    wave_sync();                              // all reads of `in` precede the writes of T below
    // ---- layout A: fft4 / fft8 leaves and pass16 ----
    {
        const float *c16 = L.cos + TB_COS16;
        const float sqrthalf = c16[2];
        const bool is16 = ((eight ? kMask16_64 >> (hl & 3) : kMask16_512 >> hl) & 1) != 0;
        sr_fft4(z[0], z[1], z[2], z[3]);
        sr_fft8_tail(z, sqrthalf);
        sr_fft4(z[8], z[9], z[10], z[11]);
        if (is16) {
            sr_fft4(z[12], z[13], z[14], z[15]);
            sr_transform_zero(z[0], z[4], z[8], z[12]);
            sr_transform(z[1], z[5], z[9], z[13], c16[1], c16[3]);
            sr_transform(z[2], z[6], z[10], z[14], sqrthalf, sqrthalf);
            sr_transform(z[3], z[7], z[11], z[15], c16[3], c16[1]);
        } else {
            sr_fft8_tail(z + 8, sqrthalf);
        }
    }
    // ---- A -> B through LDS: element e at slot e + (e >> 4) ----
#pragma unroll
    for (int i = 0; i < 16; i++) T[17 * hl + i] = z[i];
    wave_sync();
#pragma unroll
    for (int m = 0; m < 16; m++) z[m] = T[68 * c + r + 4 * m + (m >> 2)];
    wave_sync();
    // ---- layout B: pass32 on the first half, then pass64 or pass32 on the second half ----
    {
        const float *c32 = L.cos + TB_COS32, *c64 = L.cos + TB_COS64;
        const bool is64 = eight || ((kMask64_512 >> c) & 1) != 0;
        const float w32[2][2] = { { c32[r], c32[8 - r] }, { c32[r + 4], c32[4 - r] } };
#pragma unroll
        for (int kap = 0; kap < 2; kap++)
            sr_transform_sel(z[kap], z[kap + 2], z[kap + 4], z[kap + 6], w32[kap][0], w32[kap][1],
                             kap == 0 && r == 0);
        if (is64) {
#pragma unroll
            for (int kap = 0; kap < 4; kap++) {
                const int k = r + 4 * kap;
                sr_transform_sel(z[kap], z[kap + 4], z[kap + 8], z[kap + 12], c64[k], c64[16 - k],
                                 kap == 0 && r == 0);
            }
        } else {
#pragma unroll
            for (int kap = 0; kap < 2; kap++)
                sr_transform_sel(z[8 + kap], z[10 + kap], z[12 + kap], z[14 + kap], w32[kap][0], w32[kap][1],
                                 kap == 0 && r == 0);
        }
    }
    float *buf = reinterpret_cast<float *>(T);
    if (eight) {
        // ---- post-rotation of window c from layout B (mdct.c:145-158), N = 256 ----
        const float *tcos = L.rot256, *tsin = tcos + 64;
        cpx o[16];
#pragma unroll
        for (int m = 0; m < 16; m++) {
            const int e = r + 4 * m;
            cmul(o[m].re, o[m].im, z[m].im, z[m].re, tsin[e], tcos[e]);
        }
        // (every lane's layout-B reads of T are complete: the loads above were consumed)
#pragma unroll
        for (int m = 0; m < 16; m++) {
            const int e = r + 4 * m;
            buf[c * 128 + 2 * e] = o[m].re;
            buf[c * 128 + 127 - 2 * e] = o[m].im;
        }
    } else {
        // ---- B -> C through LDS: element e at slot e + 4 (e >> 6) ----
#pragma unroll
        for (int m = 0; m < 16; m++) T[68 * c + r + 4 * m] = z[m];
        wave_sync();
#pragma unroll
        for (int j = 0; j < 16; j++) z[j] = T[hl + 32 * j + 4 * (j >> 1)];
        wave_sync();
        // ---- layout C: pass128 at 0, 256, 384; pass256; pass512 ----
        const float *c128 = L.cos + TB_COS128, *c256 = L.cos + TB_COS256, *c512 = L.cos + TB_COS512;
        {
            const float wre = c128[hl], wim = c128[32 - hl];
            sr_transform_sel(z[0], z[1], z[2], z[3], wre, wim, hl == 0);
            sr_transform_sel(z[8], z[9], z[10], z[11], wre, wim, hl == 0);
            sr_transform_sel(z[12], z[13], z[14], z[15], wre, wim, hl == 0);
        }
#pragma unroll
        for (int kap = 0; kap < 2; kap++) {
            const int k = hl + 32 * kap;
            sr_transform_sel(z[kap], z[kap + 2], z[kap + 4], z[kap + 6], c256[k], c256[64 - k], kap == 0 && hl == 0);
        }
#pragma unroll
        for (int kap = 0; kap < 4; kap++) {
            const int k = hl + 32 * kap;
            sr_transform_sel(z[kap], z[kap + 4], z[kap + 8], z[kap + 12], c512[k], c512[128 - k], kap == 0 && hl == 0);
        }
        // ---- post-rotation (mdct.c:145-158): z[e] -> buf[2e] (re), buf[1023 - 2e] (im) ----
        const float *tcos = L.rot2048, *tsin = tcos + 512;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int e = hl + 32 * j;
            float re, im;
            if constexpr (TW) cmul(re, im, z[j].im, z[j].re, tw.post[j].x, tw.post[j].y);
            else              cmul(re, im, z[j].im, z[j].re, tsin[e], tcos[e]);
            buf[2 * e] = re;
            buf[1023 - 2 * e] = im;
        }
    }
    wave_sync();
}

__device__ __forceinline__ void imdct_half_regs(const CoreTabs &L, const float *in, cpx *T, bool eight, int hl)
{
    const LongTw none = {};
    imdct_half_regs_tw<false>(L, in, T, eight, hl, none);
}

// Windowing, overlap-add and the new overlap (aacdec.c:1763-1805) for one channel, whole
// wave.  buf: the channel's buf[1024] in LDS; saved in / out in HBM (may alias).
// emit(q, v) receives out[q]; within one call every lane's q is distinct and a group of
// 64 consecutive positions (ascending or descending with the lane).
template <class Emit>
__device__ __forceinline__ void core2_window(const CoreTabs &L, HeaacIcs ics, float bias, const float *buf,
                                             const float *g_saved_in, float *g_saved_out, int lane, Emit emit)
{
    const int ws0 = ics.window_sequence[0], ws1 = ics.window_sequence[1];
    const bool eight = ws0 == HEAAC_EIGHT_SHORT_SEQUENCE;
    const float *swindow      = ics.use_kb_window[0] ? L.kbd_short : L.sine_short;
    const float *lwindow_prev = ics.use_kb_window[1] ? L.kbd_long  : L.sine_long;
    const float *swindow_prev = ics.use_kb_window[1] ? L.kbd_short : L.sine_short;
    const bool long_long =
        (ws1 == HEAAC_ONLY_LONG_SEQUENCE || ws1 == HEAAC_LONG_STOP_SEQUENCE) &&
        (ws0 == HEAAC_ONLY_LONG_SEQUENCE || ws0 == HEAAC_LONG_START_SEQUENCE);

    // every read of the old overlap is issued before the new one is written (in place)
    float sv[8];
#pragma unroll
    for (int t = 0; t < 8; t++) sv[t] = g_saved_in[64 * t + lane];

    // ff_vector_fmul_window_c (dsputil.c:3832-3845), one (p, 2 len - 1 - p) pair per lane
    auto window64 = [&](int o, float a, const float *s1, const float *w, float &hi) {
        const int p = lane;
        const float b = s1[63 - p], wi = w[p], wj = w[127 - p];
        emit(o + p, a * wj - b * wi + bias);
        hi = a * wi + b * wj + bias;
    };
    if (long_long) {
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const int p = 64 * t + lane;
            const float a = sv[t], b = buf[511 - p];
            const float wi = lwindow_prev[p], wj = lwindow_prev[1023 - p];
            emit(p, a * wj - b * wi + bias);
            emit(1023 - p, a * wi + b * wj + bias);
        }
    } else {
#pragma unroll
        for (int t = 0; t < 7; t++) emit(64 * t + lane, sv[t] + bias);
        if (eight) {
            float hi;
            window64(448, sv[7], buf, swindow_prev, hi);                       emit(448 + 127 - lane, hi);
            window64(448 + 128, buf[0 * 128 + 64 + lane], buf + 128, swindow, hi);   emit(448 + 128 + 127 - lane, hi);
            window64(448 + 256, buf[1 * 128 + 64 + lane], buf + 256, swindow, hi);   emit(448 + 256 + 127 - lane, hi);
            window64(448 + 384, buf[2 * 128 + 64 + lane], buf + 384, swindow, hi);   emit(448 + 384 + 127 - lane, hi);
            // temp[0..127] of the reference: first half -> out[960..1023], second half (minus
            // bias) -> saved[0..63]
            window64(960, buf[3 * 128 + 64 + lane], buf + 512, swindow, hi);
            g_saved_out[63 - lane] = hi - bias;
        } else {
            float hi;
            window64(448, sv[7], buf, swindow_prev, hi);
            emit(448 + 127 - lane, hi);
#pragma unroll
            for (int t = 9; t < 16; t++) emit(64 * t + lane, buf[64 * t + lane - 512] + bias);
        }
    }
    // buffer update (aacdec.c:1793-1805), straight to HBM
    if (eight) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int p = lane;
            const float a = buf[(4 + j) * 128 + 64 + p], b = buf[(5 + j) * 128 + 63 - p];
            const float wi = swindow[p], wj = swindow[127 - p];
            g_saved_out[64 + 128 * j + p]       = a * wj - b * wi + 0.0f;
            g_saved_out[64 + 128 * j + 127 - p] = a * wi + b * wj + 0.0f;
        }
        g_saved_out[448 + lane] = buf[7 * 128 + 64 + lane];
    } else if (ws0 == HEAAC_LONG_START_SEQUENCE) {
#pragma unroll
        for (int t = 0; t < 7; t++) g_saved_out[64 * t + lane] = buf[512 + 64 * t + lane];
        g_saved_out[448 + lane] = buf[7 * 128 + 64 + lane];
    } else {
#pragma unroll
        for (int t = 0; t < 8; t++) g_saved_out[64 * t + lane] = buf[512 + 64 * t + lane];
    }
}

// Coefficients of one channel HBM -> LDS with the whole wave (4 KiB, 16-byte loads).
__device__ __forceinline__ void core2_stage_coeffs(float *dst, const float *g_coeffs, int lane)
{
    const float4 *c4 = reinterpret_cast<const float4 *>(g_coeffs);
    float4 *d4 = reinterpret_cast<float4 *>(dst);
#pragma unroll
    for (int i = 0; i < 4; i++) d4[lane + 64 * i] = c4[lane + 64 * i];
}
