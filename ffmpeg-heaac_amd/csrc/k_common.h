// k_common.h -- device-side building blocks shared by the gfx950 kernels.
//
// Execution model used throughout: ONE WAVEFRONT (64 lanes) OWNS ONE UNIT of
// work (a channel of a frame, a filterbank, ...) and keeps its working set in a
// private slice of LDS.  Waves of a workgroup share only the immutable tables
// staged into LDS once per (persistent) workgroup, so the per-unit code needs
// no s_barrier at all -- LDS operations of one wave are executed in issue
// order, and wave_sync() only stops the compiler from reordering them.
//
// Arithmetic is kept in the reference decoder's operation order (no FMA
// contraction: build with -ffp-contract=off); parallelism comes only from
// independent outputs, never from re-associating a sum.
#pragma once
#include <hip/hip_runtime.h>
#include "heaac_dsp.h"
#include <stdint.h>
#include "tables.h"

#define WAVE 64

// Compiler-only ordering point between LDS phases of one wave.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------
// Buffer addressing for per-unit HBM records.  A unit's record starts at a
// wave-uniform address, so every access is  base(SGPR x4) + lane part (one
// 32-bit VGPR) + uniform part (SGPR page + 12-bit immediate): no 64-bit
// per-access VGPR addresses, which the compiler otherwise hoists out of the
// unrolled slot loops and spills.  Indices are in 32-bit words.
// ---------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));

// One channel of the X hand-over workspace between the HF / PS kernels and the synthesis kernels: [38 slots][64 bands]
// [re, im] floats (slots 32..37: the hybrid filters' look-ahead).  A frame's record is two of them (left / mono, right).
// (Re and im side by side: a lane's value leaves as ONE 8-byte store, a row of 48 bands is exactly three 128-byte lines.)
#define HE_X_CHANNEL (38 * 64 * 2)

template <int ST_AUX>
struct GBufT {
    __amdgpu_buffer_rsrc_t r;
    __device__ __forceinline__ explicit GBufT(const void *base)
        : r(__builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7fffffff, 0x00020000)) {}
    // v: lane-dependent word index, s: wave-uniform word index
    __device__ __forceinline__ float ld(int v, int s = 0) const
    {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
            r, v * 4 + ((s * 4) & 4095), (s * 4) & ~4095, 0));
    }
    __device__ __forceinline__ void st(float x, int v, int s = 0) const
    {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), r,
                                              v * 4 + ((s * 4) & 4095), (s * 4) & ~4095, ST_AUX);
    }
    // same with the lane part given as a BYTE offset (hot loops keep it in one VGPR)
    __device__ __forceinline__ float ldb(int vb, int s = 0) const
    {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
            r, vb + ((s * 4) & 4095), (s * 4) & ~4095, 0));
    }
    __device__ __forceinline__ void stb(float x, int vb, int s = 0) const
    {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), r,
                                              vb + ((s * 4) & 4095), (s * 4) & ~4095, ST_AUX);
    }
    // (re, im) pair at an 8-byte aligned byte offset
    typedef unsigned int u32x2 __attribute__((vector_size(8)));
    __device__ __forceinline__ v2f ldb2(int vb, int s = 0) const
    {
        const u32x2 r2 = __builtin_amdgcn_raw_buffer_load_b64(r, vb + ((s * 4) & 4095), (s * 4) & ~4095, 0);
        return __builtin_bit_cast(v2f, r2);
    }
    __device__ __forceinline__ void stb2(v2f x, int vb, int s = 0) const
    {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, x), r,
                                              vb + ((s * 4) & 4095), (s * 4) & ~4095, ST_AUX);
    }
};
typedef GBufT<0> GBuf;

// Redefine a lane offset opaquely: address arithmetic on it cannot be hoisted out of
// an unrolled loop (where it would occupy a VGPR per access) and folds into the
// instruction's immediate offset instead.
__device__ __forceinline__ int opaque(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}

// ---------------------------------------------------------------------------
// Frame feed of a persistent wave.  Frames cost between ~0.8x and ~1.3x the mean (envelope counts, window
// sequences, DRAM jitter), and a fixed stride also locks the waves' memory traffic in step (measured again in
// round 3: k_hfps +20 % with 7/8 of the frames by stride), so the waves draw their frames from a queue word.  One
// atomic per frame runs into the word itself: 262 144 frames in 3.4 ms are 77 M atomics/s on one address, close to
// what one address serves (~88 M/s), and the tickets then arrive late.  So a ticket is worth Q consecutive frames
// (k_synth: Q = 2, -4 ... -10 % by box).  The first chunk is the wave's own index; the ticket for the chunk after
// next is issued at the top of a chunk's first frame and consumed Q frames later.  Everything here is wave-uniform;
// the ticket travels in lane 0's VGPR until it is consumed.
// ---------------------------------------------------------------------------
template <unsigned Q>
struct FrameFeed {
    unsigned long long cur, nxt;          // the frame in work and the one after it (>= n: none)
    unsigned long long base1;             // first frame of the next chunk
    unsigned long long qoff;              // frames handed out without the queue (the first chunk of every wave)
    unsigned *queue;
    unsigned pos;                         // index of `cur` in its chunk
    unsigned vt;                          // ticket in flight (lane 0)

    // wave: index of this wave in the grid, waves: waves in the grid
    __device__ __forceinline__ void init(unsigned long long wave, unsigned long long waves, unsigned *q, int lane)
    {
        queue = q;
        qoff = waves * Q;
        pos = 0;
        cur = wave * Q;
        unsigned t = 0;
        if (lane == 0) t = atomicAdd(queue, Q);
        base1 = qoff + (unsigned long long)__builtin_amdgcn_readfirstlane(t);
        nxt = Q > 1 ? cur + 1 : base1;
        vt = 0;
    }
    // call at the top of a frame
    __device__ __forceinline__ void request(int lane)
    {
        if (pos == 0) { vt = 0; if (lane == 0) vt = atomicAdd(queue, Q); }
    }
    // call at the end of a frame
    __device__ __forceinline__ void advance()
    {
        cur = nxt;
        if (pos + 1 < Q) {
            pos++;
        } else {
            pos = 0;
            base1 = qoff + (unsigned long long)__builtin_amdgcn_readfirstlane(vt);
        }
        // the frame after the new `cur`: its neighbour inside the chunk, or the next chunk's first (pos counts
        // the frames of the chunk `cur` is in; at this point base1 is the chunk after cur's)
        nxt = pos + 1 < Q ? cur + 1 : base1;
    }
};

struct NoHook {
    __device__ __forceinline__ void operator()() const {}
    __device__ __forceinline__ void operator()(int) const {}
};

// ---------------------------------------------------------------------------
// L2 prefetch of a record the wave will read a little later: every lane loads one dword of a
// different 128-byte line (sc1: past the CU's vector L1, allocated in the XCD's L2) and the data is
// thrown away.  The loads are LDS-DMA (`global_load_lds_dword`): they have NO register destination,
// so a load returning late cannot land in a register the compiler has meanwhile given to something
// else (it schedules an asm statement as one opaque instruction and knows nothing of the load in
// flight).  They write 256 bytes of LDS nobody reads: `dump` is that area's LDS byte address (wave
// uniform; all waves of a workgroup may share it).  The compiler's own vmcnt bookkeeping stays
// correct: these loads are older than anything it waits for afterwards (vmcnt retires in order).
// A kernel that touches must end with l2_touch_drain(): its LDS may go to another workgroup only
// once every DMA write has landed.
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    // low half of a flat LDS address = the LDS byte offset
    return __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<uintptr_t>(p));
}
#define L2_TOUCH_STRIDE 128        // one load per 128-byte line (a second one per line fetches nothing more)
__device__ __forceinline__ void l2_touch(const void *base, unsigned bytes, int lane, unsigned dump)
{
    for (unsigned o = 0; o < bytes; o += WAVE * L2_TOUCH_STRIDE) {
        unsigned off = o + (unsigned)lane * L2_TOUCH_STRIDE;
        off = off < bytes ? off : bytes - 4;                 // surplus lanes re-touch the last line
        const char *p = reinterpret_cast<const char *>(base) + off;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(p), "s"(dump) : "memory");
    }
}
__device__ __forceinline__ void l2_touch_drain()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------
// Split-radix schedule.  The reference FFT (libavcodec/fft.c:283-351) is the
// recursion  fft(n, o) = fft(n/2, o); fft(n/4, o+n/2); fft(n/4, o+3n/4);
// pass(n, o)  bottoming out in fft4/fft8.  A block of size M exists at offset
// o iff the bits of o/M (MSB first) parse as tokens {0, 10, 11}.  Blocks of
// one size are disjoint and only depend on smaller blocks, so all blocks of a
// size run concurrently: level order 4, 8(tail), 16, 32, ... n.
// ---------------------------------------------------------------------------
struct SrSchedule {
    // offsets (in complex elements) of blocks, by log2(size): 2 -> fft4 ... 9 -> 512
    uint16_t off[10][88];
    uint16_t cnt[10];
};

constexpr void sr_collect(SrSchedule &s, int n, int o, int bits)
{
    // fft4 blocks run for every size-4 block AND as the head of every fft8
    if (n == 4) { s.off[2][s.cnt[2]++] = (uint16_t)o; return; }
    if (n == 8) {
        s.off[2][s.cnt[2]++] = (uint16_t)o;       // fft8 starts with fft4(z)
        s.off[3][s.cnt[3]++] = (uint16_t)o;
        return;
    }
    sr_collect(s, n / 2, o, bits - 1);
    sr_collect(s, n / 4, o + n / 2, bits - 2);
    sr_collect(s, n / 4, o + 3 * (n / 4), bits - 2);
    s.off[bits][s.cnt[bits]++] = (uint16_t)o;
}

constexpr SrSchedule sr_make(int bits)
{
    SrSchedule s{};
    sr_collect(s, 1 << bits, 0, bits);
    return s;
}

// ---------------------------------------------------------------------------
// Butterflies, identical expression shapes to fft.c:213-254.
// ---------------------------------------------------------------------------
struct cpx { float re, im; };

// (re, im) in an aligned VGPR pair: v_pk_mul_f32 / v_pk_add_f32 issue at the rate of
// the scalar forms on CDNA3/4, and broadcasts / swaps / sign flips ride on their op_sel
// and neg modifiers.  Used where the data layout keeps pairs together end to end.
__device__ __forceinline__ v2f bc(float s) { return v2f{s, s}; }
__device__ __forceinline__ v2f rot90(v2f a) { return v2f{-a.y, a.x}; }     // multiplication by i

// TRANSFORM(a0,a1,a2,a3,wre,wim) + BUTTERFLIES
__device__ __forceinline__ void sr_transform(cpx &a0, cpx &a1, cpx &a2, cpx &a3,
                                             float wre, float wim)
{
    float t1 = a2.re * wre + a2.im * wim;
    float t2 = a2.im * wre - a2.re * wim;
    float t5 = a3.re * wre - a3.im * wim;
    float t6 = a3.im * wre + a3.re * wim;
    float t3 = t5 - t1;  t5 = t5 + t1;
    a2.re = a0.re - t5;  a0.re = a0.re + t5;
    a3.im = a1.im - t3;  a1.im = a1.im + t3;
    float t4 = t2 - t6;  t6 = t2 + t6;
    a3.re = a1.re - t4;  a1.re = a1.re + t4;
    a2.im = a0.im - t6;  a0.im = a0.im + t6;
}

// TRANSFORM_ZERO + BUTTERFLIES
__device__ __forceinline__ void sr_transform_zero(cpx &a0, cpx &a1, cpx &a2, cpx &a3)
{
    float t1 = a2.re, t2 = a2.im, t5 = a3.re, t6 = a3.im;
    float t3 = t5 - t1;  t5 = t5 + t1;
    a2.re = a0.re - t5;  a0.re = a0.re + t5;
    a3.im = a1.im - t3;  a1.im = a1.im + t3;
    float t4 = t2 - t6;  t6 = t2 + t6;
    a3.re = a1.re - t4;  a1.re = a1.re + t4;
    a2.im = a0.im - t6;  a0.im = a0.im + t6;
}

// fft4, fft.c:292-304
__device__ __forceinline__ void sr_fft4(cpx &z0, cpx &z1, cpx &z2, cpx &z3)
{
    float t3 = z0.re - z1.re, t1 = z0.re + z1.re;
    float t8 = z3.re - z2.re, t6 = z3.re + z2.re;
    float t4 = z0.im - z1.im, t2 = z0.im + z1.im;
    float t7 = z2.im - z3.im, t5 = z2.im + z3.im;
    z2.re = t1 - t6;  z0.re = t1 + t6;
    z3.im = t4 - t8;  z1.im = t4 + t8;
    z3.re = t3 - t7;  z1.re = t3 + t7;
    z2.im = t2 - t5;  z0.im = t2 + t5;
}

// the part of fft8 after its fft4(z), fft.c:312-323 (z[0..3] already done)
__device__ __forceinline__ void sr_fft8_tail(cpx *z, float sqrthalf)
{
    float t1 = z[4].re + z[5].re;  z[5].re = z[4].re - z[5].re;
    float t2 = z[4].im + z[5].im;  z[5].im = z[4].im - z[5].im;
    float t3 = z[6].re + z[7].re;  z[7].re = z[6].re - z[7].re;
    float t4 = z[6].im + z[7].im;  z[7].im = z[6].im - z[7].im;
    float t8 = t3 - t1;  t1 = t3 + t1;
    float t7 = t2 - t4;  t2 = t2 + t4;
    z[4].re = z[0].re - t1;  z[0].re = z[0].re + t1;
    z[4].im = z[0].im - t2;  z[0].im = z[0].im + t2;
    z[6].re = z[2].re - t7;  z[2].re = z[2].re + t7;
    z[6].im = z[2].im - t8;  z[2].im = z[2].im + t8;
    sr_transform(z[1], z[3], z[5], z[7], sqrthalf, sqrthalf);
}

// CMUL of mdct.c:108-116: p = a * b with two roundings per product
__device__ __forceinline__ void cmul(float &pre, float &pim, float are, float aim, float bre, float bim)
{
    pre = are * bre - aim * bim;
    pim = are * bim + aim * bre;
}

// ---------------------------------------------------------------------------
// Per-lane register FFT of 32 points (used by the 128-point IMDCTs of the SBR
// filterbanks: one lane = one transform).  Fully unrolled at compile time so
// z[] lives in VGPRs.  cosNN are the ff_cos tables (in LDS or SGPR-uniform).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void fft8_reg(cpx *z, float sqrthalf)
{
    sr_fft4(z[0], z[1], z[2], z[3]);
    sr_fft8_tail(z, sqrthalf);
}

__device__ __forceinline__ void fft16_reg(cpx *z, const float *c16)
{
    // fft.c:327-339
    const float sqrthalf = c16[2];
    fft8_reg(z, sqrthalf);
    sr_fft4(z[8], z[9], z[10], z[11]);
    sr_fft4(z[12], z[13], z[14], z[15]);
    sr_transform_zero(z[0], z[4], z[8], z[12]);
    sr_transform(z[2], z[6], z[10], z[14], sqrthalf, sqrthalf);
    sr_transform(z[1], z[5], z[9], z[13], c16[1], c16[3]);
    sr_transform(z[3], z[7], z[11], z[15], c16[3], c16[1]);
}

__device__ __forceinline__ void fft32_reg(cpx *z, const float *c16, const float *c32)
{
    // DECL_FFT(32,16,8), fft.c:283-290,343
    fft16_reg(z, c16);
    fft8_reg(z + 16, c16[2]);
    fft8_reg(z + 24, c16[2]);
    sr_transform_zero(z[0], z[8], z[16], z[24]);
#pragma unroll
    for (int k = 1; k < 8; k++)
        sr_transform(z[k], z[k + 8], z[k + 16], z[k + 24], c32[k], c32[8 - k]);
}

// Split-radix input permutation for n = 32 (fft.c:56-65,121-122), constexpr so
// that register indices are static.
constexpr int sr_index_c(int i, int n)
{
    if (n <= 2) return i & 1;
    int half = n >> 1, quarter = n >> 2;
    if (!(i & half)) return sr_index_c(i, half) * 2;
    return sr_index_c(i, quarter) * 4 + ((i & quarter) ? -1 : 1);
}
struct Rev32 { int v[32]; };
constexpr Rev32 make_rev32()
{
    Rev32 r{};
    for (int i = 0; i < 32; i++)
        r.v[(-sr_index_c(i, 32)) & 31] = i;
    return r;
}

// ff_imdct_half for N = 128 entirely in one lane's registers (mdct.c:124-159).
//   in(i)  : functor returning input sample i (0..63)
//   out[64]: result (z as interleaved re,im)
//   rot    : tcos[32] followed by tsin[32]
template <class In>
__device__ __forceinline__ void imdct128_reg(In in, float *out, const float *rot,
                                             const float *c16, const float *c32)
{
    constexpr Rev32 R = make_rev32();
    cpx z[32];
#pragma unroll
    for (int k = 0; k < 32; k++) {
        const int j = R.v[k];
        cmul(z[j].re, z[j].im, in(63 - 2 * k), in(2 * k), rot[k], rot[32 + k]);
    }
    fft32_reg(z, c16, c32);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        float r0, i0, r1, i1;
        cmul(r0, i1, z[15 - k].im, z[15 - k].re, rot[32 + 15 - k], rot[15 - k]);
        cmul(r1, i0, z[16 + k].im, z[16 + k].re, rot[32 + 16 + k], rot[16 + k]);
        z[15 - k].re = r0;  z[15 - k].im = i0;
        z[16 + k].re = r1;  z[16 + k].im = i1;
    }
#pragma unroll
    for (int k = 0; k < 32; k++) {
        out[2 * k]     = z[k].re;
        out[2 * k + 1] = z[k].im;
    }
}

// float_to_int16_one, dsputil.c:3972-3981
__device__ __forceinline__ int float_to_int16_one(float f)
{
    int tmp = __float_as_int(f);
    // (unsigned subtraction: for a negative float the reference's int expression overflows; the
    // two's-complement wrap is the value a sub / sar pair gives)
    if (tmp & 0xf0000)
        tmp = (int)(0x43c0ffffu - (unsigned)tmp) >> 31;
    return (int)(short)(tmp - 0x8000);
}

// float_to_int16_sse2 (x86/dsputil_mmx.c:2356-2372): cvtps2dq rounds to nearest even and answers NaN or
// |f| >= 2^31 with 0x80000000; packssdw saturates to int16.
__device__ __forceinline__ int float_to_int16_sse2(float f)
{
    const int v = fabsf(f) < 2147483648.0f ? (int)rintf(f) : (int)0x80000000;
    return v < -32768 ? -32768 : v > 32767 ? 32767 : v;
}
template <int FMT>
__device__ __forceinline__ int pcm_int16(float f)
{
    return FMT == HEAAC_PCM_S16_INTERLEAVED_SSE2 ? float_to_int16_sse2(f) : float_to_int16_one(f);
}

// Copy `count` floats global -> LDS with the whole workgroup (count % 4 == 0,
// both 16-byte aligned).
__device__ __forceinline__ void wg_copy_f4(float *dst, const float *src, int count)
{
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    float4 *d4 = reinterpret_cast<float4 *>(dst);
    for (int i = threadIdx.x; i < count / 4; i += blockDim.x)
        d4[i] = s4[i];
}
