// k_core.h -- AAC core synthesis for one channel by one wavefront:
// IMDCT (2048 or 8 x 256) + windowing + overlap-add = imdct_and_windowing(),
// libavcodec/aacdec.c:1741-1806, on LDS-resident data.
#pragma once
#include "k_common.h"
#include "heaac_dsp.h"

// Tables the core stage needs, staged in LDS once per workgroup.
// Blob range [TB_COS16, TB_SINE_SHORT + 128) is copied verbatim, so blob
// offsets index it directly.
#define CORE_TAB_FLOATS (TB_SINE_SHORT + 128)      // 3860 floats = 15440 B

struct CoreLds {
    float tab[CORE_TAB_FLOATS];
    uint16_t rev512[512];
    uint16_t rev64[64];
    uint16_t sched512[10][88];
    uint16_t sched64[7][12];
};

__device__ constexpr SrSchedule kSched512 = sr_make(9);
__device__ constexpr SrSchedule kSched64  = sr_make(6);

// Stage tables (whole workgroup, followed by __syncthreads by the caller).
__device__ __forceinline__ void core_lds_init(CoreLds &L, const float *g_tab, const uint16_t *g_rev)
{
    wg_copy_f4(L.tab, g_tab, CORE_TAB_FLOATS);
    for (int i = threadIdx.x; i < 512; i += blockDim.x)
        L.rev512[i] = g_rev[RV_512 + i];
    for (int i = threadIdx.x; i < 64; i += blockDim.x)
        L.rev64[i] = g_rev[RV_64 + i];
    for (int i = threadIdx.x; i < 10 * 88; i += blockDim.x)
        L.sched512[i / 88][i % 88] = kSched512.off[i / 88][i % 88];
    for (int i = threadIdx.x; i < 7 * 12; i += blockDim.x)
        L.sched64[i / 12][i % 12] = kSched64.off[i / 12][i % 12];
}

// One pass level (size M = 1 << bits) over `nblk` blocks per window and `nwin`
// windows of `wstride` complex elements (fft.c:257-281).
template <int BITS>
__device__ __forceinline__ void lds_pass(cpx *z, const float *cosM, const uint16_t *blk,
                                         int nblk, int nwin, int wstride, int lane)
{
    constexpr int Q = (1 << BITS) / 4;
    const int per_win = nblk * Q;
    const int total = per_win * nwin;
    for (int t = lane; t < total; t += WAVE) {
        const int w = t / per_win, r = t - w * per_win;
        const int b = r / Q, k = r & (Q - 1);
        cpx *p = z + w * wstride + blk[b] + k;
        cpx a0 = p[0], a1 = p[Q], a2 = p[2 * Q], a3 = p[3 * Q];
        if (k == 0)
            sr_transform_zero(a0, a1, a2, a3);
        else
            sr_transform(a0, a1, a2, a3, cosM[k], cosM[Q - k]);
        p[0] = a0; p[Q] = a1; p[2 * Q] = a2; p[3 * Q] = a3;
    }
    wave_sync();
}

__device__ __forceinline__ void lds_leaves(cpx *z, const uint16_t *blk4, int n4,
                                           const uint16_t *blk8, int n8,
                                           int nwin, int wstride, float sqrthalf, int lane)
{
    for (int t = lane; t < n4 * nwin; t += WAVE) {
        const int w = t / n4, b = t - w * n4;
        cpx *p = z + w * wstride + blk4[b];
        cpx z0 = p[0], z1 = p[1], z2 = p[2], z3 = p[3];
        sr_fft4(z0, z1, z2, z3);
        p[0] = z0; p[1] = z1; p[2] = z2; p[3] = z3;
    }
    wave_sync();
    for (int t = lane; t < n8 * nwin; t += WAVE) {
        const int w = t / n8, b = t - w * n8;
        cpx *p = z + w * wstride + blk8[b];
        cpx v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = p[i];
        sr_fft8_tail(v, sqrthalf);
#pragma unroll
        for (int i = 0; i < 8; i++) p[i] = v[i];
    }
    wave_sync();
}

// ff_imdct_half (mdct.c:124-159) N = 2048: in[1024] (LDS) -> zf[1024] (LDS).
__device__ __forceinline__ void imdct2048_lds(const CoreLds &L, const float *in, float *zf, int lane)
{
    cpx *z = reinterpret_cast<cpx *>(zf);
    const float *tcos = L.tab + TB_ROT2048, *tsin = tcos + 512;
    for (int k = lane; k < 512; k += WAVE) {
        cpx v;
        cmul(v.re, v.im, in[1023 - 2 * k], in[2 * k], tcos[k], tsin[k]);
        z[L.rev512[k]] = v;
    }
    wave_sync();
    const float sqrthalf = L.tab[TB_COS16 + 2];
    lds_leaves(z, L.sched512[2], kSched512.cnt[2], L.sched512[3], kSched512.cnt[3], 1, 0, sqrthalf, lane);
    lds_pass<4>(z, L.tab + TB_COS16,  L.sched512[4], kSched512.cnt[4], 1, 0, lane);
    lds_pass<5>(z, L.tab + TB_COS32,  L.sched512[5], kSched512.cnt[5], 1, 0, lane);
    lds_pass<6>(z, L.tab + TB_COS64,  L.sched512[6], kSched512.cnt[6], 1, 0, lane);
    lds_pass<7>(z, L.tab + TB_COS128, L.sched512[7], kSched512.cnt[7], 1, 0, lane);
    lds_pass<8>(z, L.tab + TB_COS256, L.sched512[8], kSched512.cnt[8], 1, 0, lane);
    lds_pass<9>(z, L.tab + TB_COS512, L.sched512[9], kSched512.cnt[9], 1, 0, lane);
    for (int k = lane; k < 256; k += WAVE) {
        cpx lo = z[255 - k], hi = z[256 + k];
        float r0, i0, r1, i1;
        cmul(r0, i1, lo.im, lo.re, tsin[255 - k], tcos[255 - k]);
        cmul(r1, i0, hi.im, hi.re, tsin[256 + k], tcos[256 + k]);
        z[255 - k] = cpx{r0, i0};
        z[256 + k] = cpx{r1, i1};
    }
    wave_sync();
}

// 8 x ff_imdct_half N = 256 (aacdec.c:1760-1761): in[8][128] -> zf[8][128].
__device__ __forceinline__ void imdct256x8_lds(const CoreLds &L, const float *in, float *zf, int lane)
{
    cpx *z = reinterpret_cast<cpx *>(zf);
    const float *tcos = L.tab + TB_ROT256, *tsin = tcos + 64;
    for (int t = lane; t < 512; t += WAVE) {
        const int w = t >> 6, k = t & 63;
        const float *x = in + w * 128;
        cpx v;
        cmul(v.re, v.im, x[127 - 2 * k], x[2 * k], tcos[k], tsin[k]);
        z[w * 64 + L.rev64[k]] = v;
    }
    wave_sync();
    const float sqrthalf = L.tab[TB_COS16 + 2];
    lds_leaves(z, L.sched64[2], kSched64.cnt[2], L.sched64[3], kSched64.cnt[3], 8, 64, sqrthalf, lane);
    lds_pass<4>(z, L.tab + TB_COS16, L.sched64[4], kSched64.cnt[4], 8, 64, lane);
    lds_pass<5>(z, L.tab + TB_COS32, L.sched64[5], kSched64.cnt[5], 8, 64, lane);
    lds_pass<6>(z, L.tab + TB_COS64, L.sched64[6], kSched64.cnt[6], 8, 64, lane);
    for (int t = lane; t < 256; t += WAVE) {
        const int w = t >> 5, k = t & 31;
        cpx *zw = z + w * 64;
        cpx lo = zw[31 - k], hi = zw[32 + k];
        float r0, i0, r1, i1;
        cmul(r0, i1, lo.im, lo.re, tsin[31 - k], tcos[31 - k]);
        cmul(r1, i0, hi.im, hi.re, tsin[32 + k], tcos[32 + k]);
        zw[31 - k] = cpx{r0, i0};
        zw[32 + k] = cpx{r1, i1};
    }
    wave_sync();
}

// ff_vector_fmul_window_c (dsputil.c:3832-3845) spread over the wave:
//   dst[p]         = s0[p]*w[2len-1-p] - s1[len-1-p]*w[p] + bias
//   dst[2len-1-p]  = s0[p]*w[p]        + s1[len-1-p]*w[2len-1-p] + bias   p in [0,len)
__device__ __forceinline__ void fmul_window_wave(float *dst, const float *s0, const float *s1,
                                                 const float *w, float bias, int len, int lane)
{
    for (int p = lane; p < len; p += WAVE) {
        const float a = s0[p], b = s1[len - 1 - p];
        const float wi = w[p], wj = w[2 * len - 1 - p];
        dst[p]               = a * wj - b * wi + bias;
        dst[2 * len - 1 - p] = a * wi + b * wj + bias;
    }
}

// imdct_and_windowing (aacdec.c:1741-1806) for one channel.
//   g_coeffs : 1024 coefficients in HBM
//   g_saved_in / g_saved_out : 512 floats each (may alias)
//   sbuf     : 1024-float LDS scratch; on return holds out[1024]
//   zbuf     : 1024-float LDS scratch (buf[] of the reference)
// The new `saved` is written straight to HBM.
__device__ __forceinline__ void core_channel(const CoreLds &L, const float *g_coeffs,
                                             const float *g_saved_in, float *g_saved_out,
                                             HeaacIcs ics, float bias,
                                             float *sbuf, float *zbuf, float *svd, int lane)
{
    // coalesced 16-byte loads: 4 KiB coefficients, 2 KiB overlap
    {
        const float4 *c4 = reinterpret_cast<const float4 *>(g_coeffs);
        float4 *s4 = reinterpret_cast<float4 *>(sbuf);
#pragma unroll
        for (int i = 0; i < 4; i++)
            s4[lane + 64 * i] = c4[lane + 64 * i];
        const float4 *v4 = reinterpret_cast<const float4 *>(g_saved_in);
        float4 *d4 = reinterpret_cast<float4 *>(svd);
#pragma unroll
        for (int i = 0; i < 2; i++)
            d4[lane + 64 * i] = v4[lane + 64 * i];
    }
    wave_sync();

    const int ws0 = ics.window_sequence[0], ws1 = ics.window_sequence[1];
    const bool eight = ws0 == HEAAC_EIGHT_SHORT_SEQUENCE;
    if (eight)
        imdct256x8_lds(L, sbuf, zbuf, lane);
    else
        imdct2048_lds(L, sbuf, zbuf, lane);

    const float *buf = zbuf;
    float *out = sbuf;                      // coefficients are dead now
    const float *swindow      = L.tab + (ics.use_kb_window[0] ? TB_KBD_SHORT : TB_SINE_SHORT);
    const float *lwindow_prev = L.tab + (ics.use_kb_window[1] ? TB_KBD_LONG  : TB_SINE_LONG);
    const float *swindow_prev = L.tab + (ics.use_kb_window[1] ? TB_KBD_SHORT : TB_SINE_SHORT);

    const bool long_long =
        (ws1 == HEAAC_ONLY_LONG_SEQUENCE || ws1 == HEAAC_LONG_STOP_SEQUENCE) &&
        (ws0 == HEAAC_ONLY_LONG_SEQUENCE || ws0 == HEAAC_LONG_START_SEQUENCE);

    if (long_long) {
        fmul_window_wave(out, svd, buf, lwindow_prev, bias, 512, lane);
    } else {
        for (int i = lane; i < 448; i += WAVE)
            out[i] = svd[i] + bias;
        if (eight) {
            fmul_window_wave(out + 448 + 0 * 128, svd + 448,          buf + 0 * 128, swindow_prev, bias, 64, lane);
            fmul_window_wave(out + 448 + 1 * 128, buf + 0 * 128 + 64, buf + 1 * 128, swindow,      bias, 64, lane);
            fmul_window_wave(out + 448 + 2 * 128, buf + 1 * 128 + 64, buf + 2 * 128, swindow,      bias, 64, lane);
            fmul_window_wave(out + 448 + 3 * 128, buf + 2 * 128 + 64, buf + 3 * 128, swindow,      bias, 64, lane);
            // temp[0..127] of the reference: first half -> out[960..1023],
            // second half (minus bias) -> saved[0..63]
            {
                const int p = lane;        // len == 64 == WAVE
                const float a = buf[3 * 128 + 64 + p], b = buf[4 * 128 + 63 - p];
                const float wi = swindow[p], wj = swindow[127 - p];
                out[448 + 4 * 128 + p] = a * wj - b * wi + bias;
                const float hi = a * wi + b * wj + bias;     // temp[127 - p]
                g_saved_out[63 - p] = hi - bias;
            }
        } else {
            fmul_window_wave(out + 448, svd + 448, buf, swindow_prev, bias, 64, lane);
            for (int i = 576 + lane; i < 1024; i += WAVE)
                out[i] = buf[i - 512] + bias;
        }
    }

    // buffer update (aacdec.c:1793-1805), straight to HBM
    if (eight) {
        // saved[64 + 128 j + ...] = window(buf[(4+j)*128+64], buf[(5+j)*128]), bias 0
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
        for (int i = lane; i < 448; i += WAVE)
            g_saved_out[i] = buf[512 + i];
        g_saved_out[448 + lane] = buf[7 * 128 + 64 + lane];
    } else {
        const float4 *b4 = reinterpret_cast<const float4 *>(buf + 512);
        float4 *o4 = reinterpret_cast<float4 *>(g_saved_out);
#pragma unroll
        for (int i = 0; i < 2; i++)
            o4[lane + 64 * i] = b4[lane + 64 * i];
    }
    wave_sync();
}
