// k_lc.hip -- batched AAC-LC channel-element synthesis and batched IMDCT.
//
// spectral_to_sample() for an SCE/CPE without SBR (aacdec.c:1903-1933):
// imdct_and_windowing() per channel with bias 385, then (optionally)
// float_to_int16_interleave (dsputil.c:3989-4001).
//
// Persistent workgroups of 12 wavefronts; each wavefront owns two channels at a time (one per
// half-wave, FFT in registers: k_core2.h), tables live in LDS for the life of the workgroup.
// HBM traffic per frame is the algorithmic minimum: coefficients and overlap read once, PCM and
// overlap written once.  The LDS-FFT kernels further down serve the stage-level
// heaac_imdct_half_batch and the FFTContext shim.
#include "k_core.h"
#include "k_core2.h"
#include "kernels.h"

#define LC_WAVES 4

// PCM stores keep the default cache policy (non-temporal measured inside the box-to-box noise, profiles/r02_experiments.md E7)
#define LC_ST(p, v) (*(p) = (v))

struct LcWaveLds {
    float sbuf[1024];
    float zbuf[1024];
    float svd[512];
    uint16_t pcm0[1024];
};

// ---------------------------------------------------------------------------
// k_lc_decode: one wavefront = two channels at a time (the two channels of a CPE, or
// two consecutive SCE frames), FFT in registers (k_core2.h).
// ---------------------------------------------------------------------------
#define LC2_WAVES 12

struct Lc2Wave {
    cpx T[2][C2_TSTRIDE];         // per channel: coefficients, transposes, then buf[1024]
    uint16_t pcm0[1024];          // left channel of an interleaved int16 pair
};

template <int CH, int FMT>
__global__ __launch_bounds__(LC2_WAVES * WAVE)
void k_lc_decode(const float *__restrict__ g_tab, const uint16_t *__restrict__ g_rev,
                 const float *__restrict__ g_coeffs, const HeaacIcs *__restrict__ g_ics,
                 const float *g_state_in, float *g_state_out,
                 void *__restrict__ g_pcm, unsigned long long n)
{
    __shared__ Core2Lds L;
    __shared__ Lc2Wave W[LC2_WAVES];
    core2_lds_init(L, g_tab, g_rev);
    const CoreTabs LT = core2_tabs(L);
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE), lane = threadIdx.x % WAVE;
    Lc2Wave &w = W[wave];
    const unsigned long long units = (unsigned long long)n * CH;      // channels
    const unsigned long long pairs = (units + 1) / 2;

    for (unsigned long long pr = (unsigned long long)blockIdx.x * LC2_WAVES + wave; pr < pairs;
         pr += (unsigned long long)gridDim.x * LC2_WAVES) {
        const unsigned long long u0 = 2 * pr;
        const bool have1 = u0 + 1 < units;                              // uniform
        const unsigned long long u1 = have1 ? u0 + 1 : u0;
        const HeaacIcs ics0 = g_ics[u0], ics1 = g_ics[u1];
        core2_stage_coeffs(reinterpret_cast<float *>(w.T[0]), g_coeffs + u0 * 1024, lane);
        core2_stage_coeffs(reinterpret_cast<float *>(w.T[1]), g_coeffs + u1 * 1024, lane);
        wave_sync();
        {
            const int half = lane >> 5, hl = lane & 31;
            const bool eight = (half ? ics1.window_sequence[0] : ics0.window_sequence[0]) == HEAAC_EIGHT_SHORT_SEQUENCE;
            imdct_half_regs(LT, reinterpret_cast<const float *>(w.T[half]), w.T[half], eight, hl);
        }
        // add_bias: 385 for the C conversion, 0 for the SIMD configuration (aacdec.c:573-581)
        constexpr float LC_BIAS = FMT == HEAAC_PCM_S16_INTERLEAVED_SSE2 ? 0.0f : HEAAC_ADD_BIAS;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            if (c == 1 && !have1) break;
            const unsigned long long u = c ? u1 : u0;
            const HeaacIcs ics = c ? ics1 : ics0;
            const float *buf = reinterpret_cast<const float *>(w.T[c]);
            const float *sin_ = g_state_in + u * 512;
            float *sout = g_state_out + u * 512;
            if (FMT == HEAAC_PCM_F32_PLANAR) {
                float *o = reinterpret_cast<float *>(g_pcm) + u * 1024;
                core2_window(LT, ics, LC_BIAS, buf, sin_, sout, lane, [&](int q, float v) { LC_ST(o + q, v); });
            } else if (CH == 1) {
                int16_t *o = reinterpret_cast<int16_t *>(g_pcm) + u * 1024;
                core2_window(LT, ics, LC_BIAS, buf, sin_, sout, lane,
                             [&](int q, float v) { LC_ST(o + q, (int16_t)pcm_int16<FMT>(v)); });
            } else if (c == 0) {
                core2_window(LT, ics, LC_BIAS, buf, sin_, sout, lane,
                             [&](int q, float v) { w.pcm0[q] = (uint16_t)pcm_int16<FMT>(v); });
                wave_sync();
            } else {
                // float_to_int16_interleave (dsputil.c:3989-4001): L from LDS, R fresh
                uint32_t *o = reinterpret_cast<uint32_t *>(g_pcm) + (u0 / 2) * 1024;
                core2_window(LT, ics, LC_BIAS, buf, sin_, sout, lane, [&](int q, float v) {
                    LC_ST(o + q, (uint32_t)w.pcm0[q] | ((uint32_t)(pcm_int16<FMT>(v) & 0xffff) << 16));
                });
            }
        }
        wave_sync();
    }
}

// Batched ff_imdct_half for the two AAC sizes (LDS split-radix) -------------
template <int WHICH>
__global__ __launch_bounds__(LC_WAVES * WAVE)
void k_imdct_half_core(const float *__restrict__ g_tab, const uint16_t *__restrict__ g_rev,
                       float *__restrict__ g_out, const float *__restrict__ g_in,
                       unsigned long long n)
{
    __shared__ CoreLds L;
    __shared__ LcWaveLds W[LC_WAVES];
    core_lds_init(L, g_tab, g_rev);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE), lane = threadIdx.x % WAVE;
    LcWaveLds &w = W[wave];
    // WHICH 0: one 1024-sample transform per unit; WHICH 1: eight 128-sample
    // transforms per unit (tail handled by clamping the copy).
    const unsigned long long per = WHICH == 0 ? 1 : 8;
    const unsigned long long units = (n + per - 1) / per;
    for (unsigned long long u = (unsigned long long)blockIdx.x * LC_WAVES + wave; u < units;
         u += (unsigned long long)gridDim.x * LC_WAVES) {
        const unsigned long long base = u * 1024;
        const unsigned long long total = n * (WHICH == 0 ? 1024 : 128);
        for (int i = lane; i < 1024; i += WAVE)
            w.sbuf[i] = base + i < total ? g_in[base + i] : 0.0f;
        wave_sync();
        if (WHICH == 0)
            imdct2048_lds(L, w.sbuf, w.zbuf, lane);
        else
            imdct256x8_lds(L, w.sbuf, w.zbuf, lane);
        for (int i = lane; i < 1024; i += WAVE)
            if (base + i < total)
                g_out[base + i] = w.zbuf[i];
        wave_sync();
    }
}

// Batched ff_imdct_half N = 128, one transform per lane (register FFT) ------
template <int WHICH>   // 2: scale 1/64, 3: scale -2
__global__ __launch_bounds__(256)
void k_imdct_half_128(const float *__restrict__ g_tab, float *__restrict__ g_out,
                      const float *__restrict__ g_in, unsigned long long n)
{
    __shared__ float rot[64];
    __shared__ float c16[8];
    __shared__ float c32[12];
    if (threadIdx.x < 64) rot[threadIdx.x] = g_tab[(WHICH == 2 ? TB_ROT128S : TB_ROT128A) + threadIdx.x];
    if (threadIdx.x < 5) c16[threadIdx.x] = g_tab[TB_COS16 + threadIdx.x];
    if (threadIdx.x < 9) c32[threadIdx.x] = g_tab[TB_COS32 + threadIdx.x];
    __syncthreads();
    for (unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; t < n;
         t += (unsigned long long)gridDim.x * blockDim.x) {
        const float *x = g_in + t * 64;
        float o[64];
        imdct128_reg([&](int i) { return x[i]; }, o, rot, c16, c32);
        float4 *o4 = reinterpret_cast<float4 *>(g_out + t * 64);
#pragma unroll
        for (int i = 0; i < 16; i++)
            o4[i] = make_float4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
    }
}

// Batched ff_fft_calc (fft.c:364-367) for nbits 5, 6, 9: in-place on permuted input.
__device__ constexpr SrSchedule kSched32 = sr_make(5);

template <int BITS>
__global__ __launch_bounds__(LC_WAVES * WAVE)
void k_fft_calc(const float *__restrict__ g_tab, float *g_z, unsigned long long n)
{
    constexpr int N = 1 << BITS;
    __shared__ float cosb[276];
    __shared__ uint16_t sched[10][88];
    __shared__ float zb[LC_WAVES][2 * 512];
    for (int i = threadIdx.x; i < 276; i += blockDim.x) cosb[i] = g_tab[i];
    for (int i = threadIdx.x; i < 10 * 88; i += blockDim.x) {
        const SrSchedule &S = BITS == 9 ? kSched512 : BITS == 6 ? kSched64 : kSched32;
        sched[i / 88][i % 88] = S.off[i / 88][i % 88];
    }
    __syncthreads();
    constexpr SrSchedule S = BITS == 9 ? kSched512 : BITS == 6 ? kSched64 : kSched32;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE), lane = threadIdx.x % WAVE;
    cpx *z = reinterpret_cast<cpx *>(zb[wave]);
    for (unsigned long long u = (unsigned long long)blockIdx.x * LC_WAVES + wave; u < n;
         u += (unsigned long long)gridDim.x * LC_WAVES) {
        float *g = g_z + u * 2 * N;
        for (int i = lane; i < 2 * N; i += WAVE) zb[wave][i] = g[i];
        wave_sync();
        lds_leaves(z, sched[2], S.cnt[2], sched[3], S.cnt[3], 1, 0, cosb[TB_COS16 + 2], lane);
        lds_pass<4>(z, cosb + TB_COS16, sched[4], S.cnt[4], 1, 0, lane);
        lds_pass<5>(z, cosb + TB_COS32, sched[5], S.cnt[5], 1, 0, lane);
        if (BITS >= 6) lds_pass<6>(z, cosb + TB_COS64, sched[6], S.cnt[6], 1, 0, lane);
        if (BITS >= 9) {
            lds_pass<7>(z, cosb + TB_COS128, sched[7], S.cnt[7], 1, 0, lane);
            lds_pass<8>(z, cosb + TB_COS256, sched[8], S.cnt[8], 1, 0, lane);
            lds_pass<9>(z, cosb + TB_COS512, sched[9], S.cnt[9], 1, 0, lane);
        }
        for (int i = lane; i < 2 * N; i += WAVE) g[i] = zb[wave][i];
        wave_sync();
    }
}

// ff_imdct_calc's symmetry extension (mdct.c:175-178): out[n] from out[n/4 .. 3n/4)
__global__ void k_imdct_mirror(float *g_out, int n, unsigned long long count)
{
    const int n2 = n >> 1, n4 = n >> 2;
    for (unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; t < count * n4;
         t += (unsigned long long)gridDim.x * blockDim.x) {
        float *o = g_out + (t / n4) * n;
        const int k = (int)(t % n4);
        o[k] = -o[n2 - k - 1];
        o[n - k - 1] = o[n2 + k];
    }
}

// ---------------------------------------------------------------------------
static int grid_for(unsigned long long units, int per_block, int blocks_per_cu)
{
    unsigned long long g = (units + per_block - 1) / per_block;
    const unsigned long long cap = 256ull * blocks_per_cu;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" int heaac_launch_lc(const float *d_tab, const uint16_t *d_rev, int channels,
                               const float *d_coeffs, const HeaacIcs *d_ics,
                               const float *d_state_in, float *d_state_out,
                               void *d_pcm, int pcm_format, size_t n, hipStream_t s)
{
    if (n == 0) return HEAAC_OK;
    const unsigned long long pairs = ((unsigned long long)n * channels + 1) / 2;
    const int grid = grid_for(pairs, LC2_WAVES, 1);
    const dim3 b(LC2_WAVES * WAVE);
#define LAUNCH(CH, FMT) \
    hipLaunchKernelGGL((k_lc_decode<CH, FMT>), dim3(grid), b, 0, s, d_tab, d_rev, d_coeffs, d_ics, \
                       d_state_in, d_state_out, d_pcm, (unsigned long long)n)
    if (channels == 1 && pcm_format == HEAAC_PCM_F32_PLANAR) LAUNCH(1, HEAAC_PCM_F32_PLANAR);
    else if (channels == 1 && pcm_format == HEAAC_PCM_S16_INTERLEAVED) LAUNCH(1, HEAAC_PCM_S16_INTERLEAVED);
    else if (channels == 2 && pcm_format == HEAAC_PCM_F32_PLANAR) LAUNCH(2, HEAAC_PCM_F32_PLANAR);
    else if (channels == 2 && pcm_format == HEAAC_PCM_S16_INTERLEAVED) LAUNCH(2, HEAAC_PCM_S16_INTERLEAVED);
    else if (channels == 1 && pcm_format == HEAAC_PCM_S16_INTERLEAVED_SSE2) LAUNCH(1, HEAAC_PCM_S16_INTERLEAVED_SSE2);
    else if (channels == 2 && pcm_format == HEAAC_PCM_S16_INTERLEAVED_SSE2) LAUNCH(2, HEAAC_PCM_S16_INTERLEAVED_SSE2);
    else return HEAAC_ERR_ARG;
#undef LAUNCH
    return hipGetLastError() == hipSuccess ? HEAAC_OK : HEAAC_ERR_HIP;
}

extern "C" int heaac_launch_imdct_half(const float *d_tab, const uint16_t *d_rev, int which,
                                       float *d_out, const float *d_in, size_t n, hipStream_t s)
{
    if (n == 0) return HEAAC_OK;
    const unsigned long long nn = n;
    switch (which) {
    case 0:
        hipLaunchKernelGGL((k_imdct_half_core<0>), dim3(grid_for(n, LC_WAVES, 2)), dim3(LC_WAVES * WAVE),
                           0, s, d_tab, d_rev, d_out, d_in, nn);
        break;
    case 1:
        hipLaunchKernelGGL((k_imdct_half_core<1>), dim3(grid_for((n + 7) / 8, LC_WAVES, 2)),
                           dim3(LC_WAVES * WAVE), 0, s, d_tab, d_rev, d_out, d_in, nn);
        break;
    case 2:
        hipLaunchKernelGGL((k_imdct_half_128<2>), dim3(grid_for(n, 256, 4)), dim3(256), 0, s,
                           d_tab, d_out, d_in, nn);
        break;
    case 3:
        hipLaunchKernelGGL((k_imdct_half_128<3>), dim3(grid_for(n, 256, 4)), dim3(256), 0, s,
                           d_tab, d_out, d_in, nn);
        break;
    default:
        return HEAAC_ERR_ARG;
    }
    return hipGetLastError() == hipSuccess ? HEAAC_OK : HEAAC_ERR_HIP;
}

extern "C" int heaac_launch_fft_calc(const float *d_tab, int nbits, float *d_z, size_t n, hipStream_t s)
{
    if (n == 0) return HEAAC_OK;
    const dim3 g(grid_for(n, LC_WAVES, 4)), b(LC_WAVES * WAVE);
    const unsigned long long nn = n;
    if (nbits == 9)      hipLaunchKernelGGL((k_fft_calc<9>), g, b, 0, s, d_tab, d_z, nn);
    else if (nbits == 6) hipLaunchKernelGGL((k_fft_calc<6>), g, b, 0, s, d_tab, d_z, nn);
    else if (nbits == 5) hipLaunchKernelGGL((k_fft_calc<5>), g, b, 0, s, d_tab, d_z, nn);
    else return HEAAC_ERR_ARG;
    return hipGetLastError() == hipSuccess ? HEAAC_OK : HEAAC_ERR_HIP;
}

extern "C" int heaac_launch_imdct_mirror(float *d_out, int n, size_t count, hipStream_t s)
{
    if (count == 0) return HEAAC_OK;
    const unsigned long long work = (unsigned long long)count * (n >> 2);
    int grid = (int)((work + 255) / 256 > 2048 ? 2048 : (work + 255) / 256);
    hipLaunchKernelGGL(k_imdct_mirror, dim3(grid), dim3(256), 0, s, d_out, n, (unsigned long long)count);
    return hipGetLastError() == hipSuccess ? HEAAC_OK : HEAAC_ERR_HIP;
}

// ---------------------------------------------------------------------------
// k_couple: apply_independent_coupling (aacdec.c:1849-1862) over a batch, elementwise and HBM-bound:
// one lane = four consecutive samples of one frame, both target channels (the coupling element's
// samples are read once).  dest + gain * (src - bias): two roundings of the product term, then the add,
// as the reference's expression (no contraction: -ffp-contract=off).
// ---------------------------------------------------------------------------
template <int CH, bool S16>
__global__ __launch_bounds__(256)
void k_couple(float *__restrict__ g_pcm, const float *__restrict__ g_cce, const HeaacCoupling *__restrict__ g_cpl,
              int16_t *__restrict__ g_s16, unsigned long long n)
{
    const unsigned long long t = (unsigned long long)blockIdx.x * 256 + threadIdx.x;   // 256 lanes per frame
    const unsigned long long f = t >> 8;
    if (f >= n) return;
    const int q = (int)(t & 255);
    const HeaacCoupling c = g_cpl[f];
    const float4 s = *reinterpret_cast<const float4 *>(g_cce + f * 1024 + 4 * q);
    const float sv[4] = { s.x, s.y, s.z, s.w };
    float out[CH][4];
#pragma unroll
    for (int ch = 0; ch < CH; ch++) {
        float4 *d = reinterpret_cast<float4 *>(g_pcm + (f * CH + ch) * 1024 + 4 * q);
        const float4 v = *d;
        out[ch][0] = v.x; out[ch][1] = v.y; out[ch][2] = v.z; out[ch][3] = v.w;
        if (c.on[ch]) {
#pragma unroll
            for (int i = 0; i < 4; i++) out[ch][i] = out[ch][i] + c.gain[ch] * (sv[i] - HEAAC_ADD_BIAS);
            *d = make_float4(out[ch][0], out[ch][1], out[ch][2], out[ch][3]);
        }
    }
    if constexpr (S16) {
        // float_to_int16_interleave (dsputil.c:3989-4001)
        int16_t *o = g_s16 + (f * 1024 + 4 * q) * CH;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int ch = 0; ch < CH; ch++) o[i * CH + ch] = (int16_t)float_to_int16_one(out[ch][i]);
    }
}

// ---------------------------------------------------------------------------
// k_interleave: float_to_int16_interleave (dsputil.c:3989-4001) for the planes of a channel layout, elementwise and
// HBM-bound: one lane = four consecutive samples of one frame in every channel (a float4 per plane in, 8 x channels
// consecutive bytes out; neighbouring lanes write neighbouring pieces).  For up to eight channels the lane packs its
// 4 x channels samples in registers and writes them as `channels` 8-byte stores (k_interleave_packed: 24 two-byte
// stores per lane for 5.1 took 2.9 ms per 65 536 frames of 2048 samples, 0.21 of the roofline --
// profiles/r04_experiments.md E9); more channels take the sample-by-sample form.
// ---------------------------------------------------------------------------
struct PlaneArgs {
    const float *base[HEAAC_MAX_PCM_PLANES];
    unsigned long long stride[HEAAC_MAX_PCM_PLANES];
};
template <int FMT>
__global__ __launch_bounds__(256)
void k_interleave(PlaneArgs p, int channels, int len, int16_t *__restrict__ g_out, unsigned long long n)
{
    const unsigned long long quads = (unsigned long long)(len >> 2);
    const unsigned long long t = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    const unsigned long long f = t / quads;
    if (f >= n) return;
    const unsigned long long q = t - f * quads;
    int16_t *o = g_out + (f * (unsigned long long)len + 4 * q) * channels;
    for (int c = 0; c < channels; c++) {
        const float4 v = *reinterpret_cast<const float4 *>(p.base[c] + f * p.stride[c] + 4 * q);
        o[c]                = (int16_t)pcm_int16<FMT>(v.x);
        o[channels + c]     = (int16_t)pcm_int16<FMT>(v.y);
        o[2 * channels + c] = (int16_t)pcm_int16<FMT>(v.z);
        o[3 * channels + c] = (int16_t)pcm_int16<FMT>(v.w);
    }
}

template <int FMT, int CH>
__global__ __launch_bounds__(256)
void k_interleave_packed(PlaneArgs p, int len, int16_t *__restrict__ g_out, unsigned long long n)
{
    const unsigned long long quads = (unsigned long long)(len >> 2);
    const unsigned long long t = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    const unsigned long long f = t / quads;
    if (f >= n) return;
    const unsigned long long q = t - f * quads;
    // sample k of channel c is int16 number k * CH + c of the lane's 4 * CH: two to a dword, four to a store
    unsigned half[4 * CH];
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const float4 v = *reinterpret_cast<const float4 *>(p.base[c] + f * p.stride[c] + 4 * q);
        half[c]          = (unsigned)(unsigned short)(int16_t)pcm_int16<FMT>(v.x);
        half[CH + c]     = (unsigned)(unsigned short)(int16_t)pcm_int16<FMT>(v.y);
        half[2 * CH + c] = (unsigned)(unsigned short)(int16_t)pcm_int16<FMT>(v.z);
        half[3 * CH + c] = (unsigned)(unsigned short)(int16_t)pcm_int16<FMT>(v.w);
    }
    uint2 *o = reinterpret_cast<uint2 *>(g_out + (f * (unsigned long long)len + 4 * q) * CH);      // 8 * CH bytes per lane: 8-byte aligned
#pragma unroll
    for (int j = 0; j < CH; j++)
        o[j] = make_uint2(half[4 * j] | (half[4 * j + 1] << 16), half[4 * j + 2] | (half[4 * j + 3] << 16));
}

template <int FMT>
static bool launch_interleave_packed(const PlaneArgs &a, int channels, int len, int16_t *d_out, size_t n, unsigned blocks, hipStream_t stream)
{
    switch (channels) {
    case 1: k_interleave_packed<FMT, 1><<<blocks, 256, 0, stream>>>(a, len, d_out, n); return true;
    case 2: k_interleave_packed<FMT, 2><<<blocks, 256, 0, stream>>>(a, len, d_out, n); return true;
    case 3: k_interleave_packed<FMT, 3><<<blocks, 256, 0, stream>>>(a, len, d_out, n); return true;
    case 4: k_interleave_packed<FMT, 4><<<blocks, 256, 0, stream>>>(a, len, d_out, n); return true;
    case 5: k_interleave_packed<FMT, 5><<<blocks, 256, 0, stream>>>(a, len, d_out, n); return true;
    case 6: k_interleave_packed<FMT, 6><<<blocks, 256, 0, stream>>>(a, len, d_out, n); return true;
    case 7: k_interleave_packed<FMT, 7><<<blocks, 256, 0, stream>>>(a, len, d_out, n); return true;
    case 8: k_interleave_packed<FMT, 8><<<blocks, 256, 0, stream>>>(a, len, d_out, n); return true;
    default: return false;
    }
}

extern "C" int heaac_launch_interleave(int channels, const HeaacPlaneRef *planes, int len, int pcm_format,
                                       int16_t *d_out, size_t n, hipStream_t stream)
{
    PlaneArgs a = {};
    for (int c = 0; c < channels; c++) { a.base[c] = planes[c].d_base; a.stride[c] = planes[c].frame_stride; }
    const unsigned long long lanes = (unsigned long long)n * (unsigned)(len >> 2);
    const unsigned long long blocks = (lanes + 255) / 256;
    if (blocks > 0x7fffffffull) return HEAAC_ERR_ARG;
    if (((uintptr_t)d_out & 7) == 0 &&                  // (the packed form writes 8-byte words)
        (pcm_format == HEAAC_PCM_S16_INTERLEAVED_SSE2
            ? launch_interleave_packed<HEAAC_PCM_S16_INTERLEAVED_SSE2>(a, channels, len, d_out, n, (unsigned)blocks, stream)
            : launch_interleave_packed<HEAAC_PCM_S16_INTERLEAVED>(a, channels, len, d_out, n, (unsigned)blocks, stream)))
        return hipGetLastError() == hipSuccess ? HEAAC_OK : HEAAC_ERR_HIP;
    if (pcm_format == HEAAC_PCM_S16_INTERLEAVED_SSE2)
        k_interleave<HEAAC_PCM_S16_INTERLEAVED_SSE2><<<(unsigned)blocks, 256, 0, stream>>>(a, channels, len, d_out, n);
    else
        k_interleave<HEAAC_PCM_S16_INTERLEAVED><<<(unsigned)blocks, 256, 0, stream>>>(a, channels, len, d_out, n);
    return hipGetLastError() == hipSuccess ? HEAAC_OK : HEAAC_ERR_HIP;
}

extern "C" int heaac_launch_couple(int channels, float *d_pcm, const float *d_cce, const HeaacCoupling *d_cpl,
                        int16_t *d_s16, size_t n, hipStream_t stream)
{
    const unsigned grid = (unsigned)n;                 // 256 lanes = one frame
    if (channels == 2) {
        if (d_s16) k_couple<2, true><<<grid, 256, 0, stream>>>(d_pcm, d_cce, d_cpl, d_s16, n);
        else       k_couple<2, false><<<grid, 256, 0, stream>>>(d_pcm, d_cce, d_cpl, d_s16, n);
    } else {
        if (d_s16) k_couple<1, true><<<grid, 256, 0, stream>>>(d_pcm, d_cce, d_cpl, d_s16, n);
        else       k_couple<1, false><<<grid, 256, 0, stream>>>(d_pcm, d_cce, d_cpl, d_s16, n);
    }
    return hipGetLastError() == hipSuccess ? HEAAC_OK : HEAAC_ERR_HIP;
}
