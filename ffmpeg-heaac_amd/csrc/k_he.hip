// k_he.hip -- HE-AAC (SBR) batched kernels for gfx950 and the HE pipeline driver.
//
//   k_core_ana  : imdct_and_windowing (bias 0) + sbr_qmf_analysis, two channels per wave   (a8, a11)
//   k_hfadj     : lf_gen, inverse filter, chirp, hf_gen, mapping, env_estimate, gain_calc,
//                 hf_assemble, x_gen (k_hf.h) for HE-AACv1; for HE-AACv2 the same stage runs
//                 fused with Parametric Stereo in k_hfps (k_ps.hip)                        (a10, a12-a19)
//   k_synth     : sbr_qmf_synthesis + float_to_int16_interleave                            (a20, a27)
//   k_qmf_analysis / k_qmf_synthesis / k_qmf_synthesis_ds : the stage-level batched filterbanks
//
// One wavefront owns one unit (an SBR channel or a pair, an output frame); workgroups are
// persistent and keep the immutable tables in LDS.  Stages hand W[32][32][2] and X[2][2][38][64]
// to each other through a workspace, one chunk of frames at a time (capi.hip).
//
// Reference line numbers are libavcodec/aacsbr.c and aacps.c.
#include <stdlib.h>
#include <type_traits>
#include "k_core.h"
#include "k_core2.h"
#include "kernels.h"

#include "k_hf.h"

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

// ---------------------------------------------------------------------------
// k_core_ana: one wavefront = two SBR channels at a time (the two channels of a CPE, or
// two consecutive mono frames).  Core IMDCT in registers (k_core2.h), then the analysis
// filterbank with one 128-point IMDCT per lane: 2 x 32 slots fill the wave.
// ---------------------------------------------------------------------------
#define CA_WAVES 7               // (6 with all transform tables in LDS: profiles/r03_experiments.md E31)
#define CA_U     2080             // fold rows u[32][65] of one channel

struct CaWave {
    float tu[2 * 2 * C2_TSTRIDE]; // core: T[2][C2_TSTRIDE] complex; afterwards u of channel 0
    float x1[1312];               // analysis input of channel 1: 288 history + 1024 new samples
    float x0u1[CA_U];             // analysis input of channel 0 (first 1312 floats); once its fold is
                                  // done, u of channel 1
};
static_assert(CA_U >= 1312, "x of channel 0 fits under u of channel 1");
static_assert(CA_U <= 2 * 2 * C2_TSTRIDE, "u rows of channel 0 lie over the core's transpose regions");

__global__ __launch_bounds__(CA_WAVES * WAVE)
void k_core_ana(const float *__restrict__ g_tab, const uint16_t *__restrict__ g_rev,
                const float *__restrict__ g_coeffs, const HeaacIcs *__restrict__ g_ics,
                const float *g_state_in, float *g_state_out, int state_words,
                int ncore, int off_saved0, int off_sbr0,
                float *__restrict__ g_W, float scale, unsigned long long n_units)
{
    // Tables: the small ones in LDS; the long windows (8 KB), the N = 2048 post-rotation table (4 KB) and the pre-rotation
    // twiddles (4.5 KB) are read from the table blob in global memory (cache resident, coalesced, loaded well ahead of
    // their use) -- the LDS they would take is what a seventh wave per CU needs.
    __shared__ float s_cos[TB_ROT2048];          // ff_cos_16 .. ff_cos_512
    __shared__ float s_rot256[128];
    __shared__ float s_wshort[256];              // KBD then sine, 128 each
    __shared__ uint16_t s_kA512[512], s_kA64[64];
    __shared__ float s_qmf_ds[320];
    __shared__ float s_rot[64];                  // SBR analysis MDCT: tcos[32], tsin[32]
    __shared__ CaWave S[CA_WAVES];
    static_assert(TB_SINE_SHORT == TB_KBD_SHORT + 128, "the two short windows are copied as one run");
    for (int i = threadIdx.x; i < TB_ROT2048; i += blockDim.x) s_cos[i] = g_tab[i];
    for (int i = threadIdx.x; i < 128; i += blockDim.x) s_rot256[i] = g_tab[TB_ROT256 + i];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_wshort[i] = g_tab[TB_KBD_SHORT + i];
    for (int k = threadIdx.x; k < 512; k += blockDim.x) {
        const int e = g_rev[RV_512 + k];
        s_kA512[(e & 15) * 32 + (e >> 4)] = (uint16_t)k;
    }
    for (int k = threadIdx.x; k < 64; k += blockDim.x) {
        const int e = g_rev[RV_64 + k];
        s_kA64[(e & 15) * 4 + (e >> 4)] = (uint16_t)k;
    }
    for (int i = threadIdx.x; i < 320; i += blockDim.x) s_qmf_ds[i] = g_tab[TB_QMF_DS + i];
    for (int i = threadIdx.x; i < 64; i += blockDim.x)  s_rot[i] = g_tab[TB_ROT128A + i];
    __syncthreads();
    const CoreTabs L = { s_cos, g_tab + TB_ROT2048, s_rot256, g_tab + TB_KBD_LONG, g_tab + TB_SINE_LONG,
                         s_wshort, s_wshort + 128,
                         reinterpret_cast<const float2 *>(g_tab + TB_ROTA512), reinterpret_cast<const float2 *>(g_tab + TB_ROTA64),
                         s_kA512, s_kA64 };
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE), lane0 = threadIdx.x % WAVE;
    CaWave &w = S[wave];
    cpx *T0 = reinterpret_cast<cpx *>(w.tu), *T1 = T0 + C2_TSTRIDE;
    const float *c16 = s_cos + TB_COS16, *c32 = s_cos + TB_COS32;
    const unsigned long long pairs = (n_units + 1) / 2;
    LongTw tw;
    core2_load_long_twiddles(L, lane0 & 31, tw);
    for (unsigned long long pr = (unsigned long long)blockIdx.x * CA_WAVES + wave; pr < pairs;
         pr += (unsigned long long)gridDim.x * CA_WAVES) {
        const unsigned long long u0 = 2 * pr;
        const bool have1 = u0 + 1 < n_units;                            // uniform
        const unsigned long long u1 = have1 ? u0 + 1 : u0;
        const HeaacIcs ics0 = g_ics[u0], ics1 = g_ics[u1];
        int lane = opaque(lane0);      // lane-derived addresses are recomputed per pair, not hoisted and spilled
        core2_stage_coeffs(reinterpret_cast<float *>(T0), g_coeffs + u0 * 1024, lane);
        core2_stage_coeffs(reinterpret_cast<float *>(T1), g_coeffs + u1 * 1024, lane);
        wave_sync();
        {
            const int half = lane >> 5, hl = lane & 31;
            const bool eight = (half ? ics1.window_sequence[0] : ics0.window_sequence[0]) == HEAAC_EIGHT_SHORT_SEQUENCE;
            cpx *T = half ? T1 : T0;
            imdct_half_regs_tw<true>(L, reinterpret_cast<const float *>(T), T, eight, hl, tw);
        }
        // windowing (bias 0) -> x = [history 288 | out * scale] per channel (vector_fmul_scalar,
        // aacsbr.c:1142), x history in / out
#pragma unroll
        for (int c = 0; c < 2; c++) {
            if (c == 1 && !have1) break;
            const unsigned long long u = c ? u1 : u0;
            const unsigned long long f = u / ncore;
            const int ch = (int)(u - f * ncore);
            const float *st_in = g_state_in + f * state_words;
            float *st_out = g_state_out + f * state_words;
            const int off_saved = off_saved0 + ch * HEAAC_ST_SAVED;
            const int off_sbr = off_sbr0 + ch * HEAAC_ST_SBR;
            float *x = c ? w.x1 : w.x0u1;
            const float *xh_in = st_in + off_sbr + HEAAC_SBR_XHIST;
            float xh[5];
#pragma unroll
            for (int t = 0; t < 5; t++) xh[t] = 64 * t + lane < 288 ? xh_in[64 * t + lane] : 0.0f;
            const float *buf = reinterpret_cast<const float *>(c ? T1 : T0);
            if (scale != 1.0f)
                core2_window(L, c ? ics1 : ics0, 0.0f, buf, st_in + off_saved, st_out + off_saved, lane,
                             [&](int q, float v) { x[288 + q] = v * scale; });
            else
                core2_window(L, c ? ics1 : ics0, 0.0f, buf, st_in + off_saved, st_out + off_saved, lane,
                             [&](int q, float v) { x[288 + q] = v; });
#pragma unroll
            for (int t = 0; t < 5; t++) if (64 * t + lane < 288) x[64 * t + lane] = xh[t];
        }
        wave_sync();
        lane = opaque(lane);
        // sbr_qmf_analysis (aacsbr.c:1136-1169).  z[n] = ds[n] * x[319 - n]; f[k] = sum of five
        // taps 64 apart: lane = k keeps its taps in registers and walks the 32 slots.
#pragma unroll
        for (int c = 0; c < 2; c++) {
            if (c == 1 && !have1) break;
            const unsigned long long u = c ? u1 : u0;
            const unsigned long long f = u / ncore;
            const int ch = (int)(u - f * ncore);
            float *xh_out = g_state_out + f * state_words + off_sbr0 + ch * HEAAC_ST_SBR + HEAAC_SBR_XHIST;
            // (channel 1's u overwrites channel 0's x: every read of it was issued by then)
            const float *x = c ? w.x1 : w.x0u1;
            float *uu = c ? w.x0u1 : w.tu;
#pragma unroll
            for (int t = 0; t < 5; t++) if (64 * t + lane < 288) xh_out[64 * t + lane] = x[1024 + 64 * t + lane];
            const int k = lane;
            const float w0 = s_qmf_ds[k], w1 = s_qmf_ds[k + 64], w2 = s_qmf_ds[k + 128],
                        w3 = s_qmf_ds[k + 192], w4 = s_qmf_ds[k + 256];
            // Two slots per packed multiply / add.  Tap t of slot i is x[32 i + 319 - k - 64 t]: the sample that is tap 0
            // of the slot pair (i, i + 1) is tap t of the pair (i + 2 t, i + 2 t + 1), so each pair reads ONE new sample pair
            // from LDS and takes the other four from the pairs before it (10 -> 2 reads per pair: the fold was a quarter
            // of this kernel's time, profiles/r03_experiments.md E28).
            const float *xk = x + 319 - k;
            v2f a1 = v2f{xk[-64], xk[-32]}, a2 = v2f{xk[-128], xk[-96]}, a3 = v2f{xk[-192], xk[-160]},
                a4 = v2f{xk[-256], xk[-224]};
#pragma unroll
            for (int i = 0; i < 32; i += 2) {
                const v2f a0 = v2f{xk[32 * i], xk[32 * i + 32]};
                const v2f f = bc(w0) * a0 + bc(w1) * a1 + bc(w2) * a2 + bc(w3) * a3 + bc(w4) * a4;
                uu[i * 65 + k] = f.x;
                uu[(i + 1) * 65 + k] = f.y;
                a4 = a3; a3 = a2; a2 = a1; a1 = a0;
            }
        }
        wave_sync();
        lane = opaque(lane);
        // shuffle to the IMDCT input (:1155-1160): in[0] = f[0]; in[2k-1] = f[k];
        // in[2k] = -f[64-k] (k = 1..31); in[63] = f[32];  then ff_imdct_half (N = 128),
        // lane = (channel, slot)
        {
            float *row = (lane >> 5 ? w.x0u1 : w.tu) + (lane & 31) * 65;
            const float *f = row;
            float o[64];
            imdct128_reg([&](int j) -> float {
                             if (j == 0)  return f[0];
                             if (j == 63) return f[32];
                             return (j & 1) ? f[(j + 1) >> 1] : -f[64 - (j >> 1)];
                         }, o, s_rot, c16, c32);
            // W[1][i][k] = (-z[63-k], z[k])                              (:1163-1166)
#pragma unroll
            for (int k = 0; k < 32; k++) {
                row[2 * k]     = -o[63 - k];
                row[2 * k + 1] = o[k];
            }
        }
        wave_sync();
        // coalesced store: 2048 floats per channel
#pragma unroll
        for (int c = 0; c < 2; c++) {
            if (c == 1 && !have1) break;
            const float *uu = c ? w.x0u1 : w.tu;
            float *Wo = g_W + (c ? u1 : u0) * 2048;
            // four rows per pass, sixteen bytes per lane: 8 stores of 1 KiB instead of 32 of 256 bytes (the LDS rows have
            // an odd stride, so their four floats are read one by one: two lanes per bank, the minimum for a wave)
#pragma unroll
            for (int it = 0; it < 8; it++) {
                const int r = 4 * it + (lane >> 4), q = 4 * (lane & 15);
                const float *sr = uu + r * 65 + q;
                *reinterpret_cast<float4 *>(Wo + r * 64 + q) = make_float4(sr[0], sr[1], sr[2], sr[3]);
            }
        }
        wave_sync();
    }
}

__global__ __launch_bounds__(HF_WAVES * WAVE)
void k_hfadj(const float *__restrict__ g_tab,
             const HeaacSbrFrame *__restrict__ g_sbr, const HeaacSbrHeader *__restrict__ g_hdr, unsigned n_hdr,
             const float *g_W, const float *g_state_in, float *g_state_out, int state_words,
             int ncore, int off_sbr0, float *g_X, unsigned long long n_units, unsigned *g_queue,
             unsigned char *__restrict__ g_xtop)
{
    __shared__ float s_xlow[HF_WAVES][HF_XLOW_WORDS], s_aux[HF_WAVES][HF_AUX_WORDS], s_rec[HF_WAVES][HF_REC_WORDS];
    __shared__ float s_noise[1024];              // sbr_noise_table, staged once per workgroup
    wg_copy_f4(s_noise, g_tab + TB_NOISE, 1024);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE), lane = threadIdx.x % WAVE;
    const HfWave S = hf_wave_view(s_xlow[wave], s_aux[wave], s_rec[wave]);
    // Units are drawn from a queue two at a time (for a CPE: the two channels of one frame, which share the
    // frame's side info): FrameFeed over pairs, one pair per ticket (k_common.h).
    FrameFeed<1> feed;
    feed.init((unsigned long long)blockIdx.x * HF_WAVES + wave, (unsigned long long)gridDim.x * HF_WAVES, g_queue, lane);
    while (feed.cur * 2 < n_units) {
        const unsigned long long ub = feed.cur * 2;
        feed.request(lane);
      for (int qi = 0; qi < 2 && ub + qi < n_units; qi++) {
        const unsigned long long u = ub + qi;
        const unsigned long long f = u / ncore;
        const int ch = (int)(u - f * ncore);
        const int off = off_sbr0 + ch * HEAAC_ST_SBR;
        v2f *Xc = reinterpret_cast<v2f *>(g_X + (f * 2 + ch) * HE_X_CHANNEL);
        int xb = 64;
        hf_channel(S, s_noise, &g_sbr[f], g_hdr, n_hdr, ch, g_W + u * 2048,
                   g_state_in + f * state_words + off, g_state_out + f * state_words + off, lane,
                   [&](int i, float re, float im) {
                       if (i == 0 && g_xtop) {            // (g_xtop == nullptr: a PS stage follows and reads every band)
                           // sbr_x_gen writes the literal +0 above kx + m (aacsbr.c:1433-1444) -- in its first i_Temp slots
                           // above the PREVIOUS frame's range (:1419-1432): if there are no such slots or that range ends
                           // inside this one, the bands from kx + m (rounded up to a 128-byte line) on are +0 in every
                           // slot.  They are not stored; the unit's byte says how many bands are, and k_synth reads the
                           // rest from its page of zeros.
                           const int top16 = (S.h.kx + S.h.m + 15) & ~15;
                           const int t_old = S.c[ch].t_env_num_env_old;
                           const bool zero = 2 * t_old - 32 <= 0 || (int)g_sbr[f].kx_old + (int)g_sbr[f].m_old <= top16;
                           xb = __builtin_amdgcn_readfirstlane(zero && top16 < 64 ? top16 : 64);
                           if (lane == 0 && xb != 64) g_xtop[f * 2 + ch] = (unsigned char)xb;
                       }
                       // written once, read by k_synth a whole batch later: non-temporal (-7 % kernel time)
                       if (lane < xb) __builtin_nontemporal_store(v2f{re, im}, Xc + i * 64 + lane);
                   });
      }
        feed.advance();
    }
}

// ===========================================================================
// K_D  QMF synthesis (64 bands, div = 0) + output, one wave per frame
// ===========================================================================
#define SYN_WAVES 6
#define VB_STRIDE 129             // v slot row: 128 + 1 pad
#define VB_ROWS   41              // 32 new slots (newest first) + 9 history slots

#define SYN_WAVES_F32 7           // k_synth (both PCM formats)
struct SynWave {
    __attribute__((aligned(16))) float vb[VB_ROWS * VB_STRIDE + 3];
};
template <int NW>
struct SynLdsT {
    float win[640];               // sbr_qmf_window_us
    float rot[64];                // SBR synthesis MDCT (scale 1/64): tcos[32], tsin[32]
    float c16[8], c32[12];
    SynWave w[NW];
};
typedef SynLdsT<SYN_WAVES> SynLds;

#ifdef HEAAC_STAMPS
#define SSTAMP(i) TL_STAMP(i, (i) == 0)
#else
#define SSTAMP(i) do {} while (0)
#endif

// swap with the neighbouring lane (lane ^ 1): DPP quad_perm [1,0,3,2]
__device__ __forceinline__ float lane_xor1(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
}

// sbr_qmf_synthesis (aacsbr.c:1175-1230), div = 0, for one channel, in four pieces so that a kernel
// can have the NEXT channel's loads in flight while the current one runs its polyphase sum:
//   syn_load     X rows (lane = (slot, re/im plane): 64 floats) and the ring state (18 floats / lane)
//   syn_rows     ring state -> v rows 32..40; 64 IMDCTs (N = 128), butterfly -> v rows 0..31
//   syn_poly     10-tap polyphase sum, lane = output column; emit(i, n, value) receives out[64 i + n]
//   syn_hist_out new ring state = v rows 0..8
struct SynIn {
    float x[64];                  // X[part][slot][0..63] of this lane's (slot, part)
    float h[18];                  // v_in[lane + 64 r]
};

// The X rows are loaded coalesced -- every instruction takes 1 KiB of consecutive memory (four rows),
// each 128-byte line is fetched once -- and reach the lane that runs the row's IMDCT through a staging
// image in the wave's v-row memory (free between two channels): syn_rows starts with that transpose.
// (A lane reading its own 256-byte row in 16-byte pieces pulls every line through the vector L1 eight
// times; with 7 waves per CU the L1 keeps none of them.)
#define SYN_STAGE_STRIDE 68       // floats per staged row: 64 + 4 (b128 reads of 64 different rows spread over all banks)
// Streamed once: X rows and ring state in -- non-temporal loads: nothing for the float output (within the noise of
// a box), but the int16 output's half-line stores find their lines still in L2 more often (k_synth<1> writes 22.6
// instead of 25.4 KiB per frame and runs 3 % shorter: profiles/r04_experiments.md E6) -- PCM and ring state out
// (non-temporal stores: -5.9 % kernel time, profiles/r02_experiments.md E7).
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 syn_ld4(const f32x4 *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ float syn_ld1(const float *p) { return __builtin_nontemporal_load(p); }
template <class T>
__device__ __forceinline__ void syn_st(T *p, T v) { __builtin_nontemporal_store(v, p); }
// PLANES = true: X0 / X1 are the re / im planes [32][64] of the stage-level entry point (heaac_qmf_synthesis_batch).
// PLANES = false: X0 is a channel of the decoders' hand-over workspace, [slot][band][re, im] (X1 unused): load q holds
// the bands 2 (lane & 31), + 1 of slot 2 q + (lane >> 5) as (re, im, re, im).
template <bool PLANES>
__device__ __forceinline__ void syn_load(const float *X0, const float *X1, const float *v_in, int lane, SynIn &d)
{
    const f32x4 *p0 = reinterpret_cast<const f32x4 *>(X0), *p1 = reinterpret_cast<const f32x4 *>(X1);
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const f32x4 t = PLANES ? syn_ld4(q < 8 ? p0 + q * 64 + lane : p1 + (q - 8) * 64 + lane)
                               : syn_ld4(p0 + q * 64 + lane);
        d.x[4 * q] = t.x; d.x[4 * q + 1] = t.y; d.x[4 * q + 2] = t.z; d.x[4 * q + 3] = t.w;
    }
#pragma unroll
    for (int r = 0; r < 18; r++) d.h[r] = syn_ld1(v_in + lane + 64 * r);
}

template <bool PLANES, class SL>
__device__ __forceinline__ void syn_rows(const SL &S, SynWave &w, const SynIn &d, int lane)
{
    // staged image: row (plane, slot) at (plane * 32 + slot) * SYN_STAGE_STRIDE.  PLANES: piece q of the loads holds
    // floats 4 (lane & 15) .. of row 4 (q & 7) + (lane >> 4) of plane q >> 3; else (re, im) of two bands of one slot: the
    // two re go to the slot's plane-0 row, the two im to its plane-1 row
    float x[64];
    {
        float4 *st = reinterpret_cast<float4 *>(w.vb);
        if constexpr (PLANES) {
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int row = (q >> 3) * 32 + 4 * (q & 7) + (lane >> 4);
                st[(row * SYN_STAGE_STRIDE >> 2) + (lane & 15)] = make_float4(d.x[4 * q], d.x[4 * q + 1], d.x[4 * q + 2], d.x[4 * q + 3]);
            }
        } else {
            float2 *st2 = reinterpret_cast<float2 *>(w.vb);
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int slot = 2 * q + (lane >> 5), pi = lane & 31;
                st2[(slot * SYN_STAGE_STRIDE >> 1) + pi] = make_float2(d.x[4 * q], d.x[4 * q + 2]);
                st2[((32 + slot) * SYN_STAGE_STRIDE >> 1) + pi] = make_float2(d.x[4 * q + 1], d.x[4 * q + 3]);
            }
        }
        wave_sync();
        const int mine = (lane & 1) * 32 + (lane >> 1);
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const float4 t = st[(mine * SYN_STAGE_STRIDE >> 2) + q];
            x[4 * q] = t.x; x[4 * q + 1] = t.y; x[4 * q + 2] = t.z; x[4 * q + 3] = t.w;
        }
        wave_sync();
    }
    SSTAMP(1);
    // history: 9 slots behind the 32 new ones
#pragma unroll
    for (int r = 0; r < 18; r++)
        w.vb[(32 + (r >> 1)) * VB_STRIDE + (r & 1) * 64 + lane] = d.h[r];
    // 64 IMDCTs (N = 128): lane = (slot, re/im plane).  X[1][i][n] = -X[1][i][n] for odd n (:1201-1203):
    // the odd lanes flip the sign bit of their odd inputs, so that all lanes run ONE instruction stream.
    const int i = lane >> 1, part = lane & 1;
    const unsigned flip = (unsigned)part << 31;
    float o[64];
    imdct128_reg([&](int j) -> float {
                     return (j & 1) ? __uint_as_float(__float_as_uint(x[j]) ^ flip) : x[j];
                 }, o, S.rot, S.c16, S.c32);
    SSTAMP(2);
    // v[n] = -buf0[63-n] + buf1[n];  v[127-n] = buf0[63-n] + buf1[n]   (:1206-1209)
    // The even lane holds buf0 and forms v[n], the odd lane holds buf1 and forms v[64 + n] =
    // buf0[n] + buf1[63 - n]: at step n both send o[n] to the partner and add their own o[63 - n]
    // (negated in the even lane), and both store at column n of their half of the row.
    float *vs = w.vb + (31 - i) * VB_STRIDE + 64 * part;
    const unsigned neg = (unsigned)(part ^ 1) << 31;
#pragma unroll
    for (int n = 0; n < 64; n++) {
        const float mine = __uint_as_float(__float_as_uint(o[63 - n]) ^ neg);
        vs[n] = lane_xor1(o[n]) + mine;
    }
}

template <int UNROLL = 1, class SL, class Emit>
__device__ __forceinline__ void syn_poly(const SL &S, const SynWave &w, float scale, float bias, int lane, Emit emit)
{
    SSTAMP(3);
    // 10-tap polyphase sum (:1210-1219), lane = n
    const int n = lane;
    float wt[10];
#pragma unroll
    for (int j = 0; j < 10; j++) wt[j] = S.win[64 * j + n];
    const bool scale_and_bias = scale != 1.0f || bias != 0.0f;
    // two slots per packed multiply / add: (slot i, slot i + 1) share the window taps
#pragma unroll UNROLL
    for (int i = 0; i < 32; i += 2) {
        const float *va = w.vb + (31 - i) * VB_STRIDE + n, *vb = va - VB_STRIDE;
        v2f acc = v2f{va[0], vb[0]} * bc(wt[0]) + v2f{0.0f, 0.0f};
        acc = v2f{va[1 * VB_STRIDE + 64], vb[1 * VB_STRIDE + 64]} * bc(wt[1]) + acc;
        acc = v2f{va[2 * VB_STRIDE],      vb[2 * VB_STRIDE]}      * bc(wt[2]) + acc;
        acc = v2f{va[3 * VB_STRIDE + 64], vb[3 * VB_STRIDE + 64]} * bc(wt[3]) + acc;
        acc = v2f{va[4 * VB_STRIDE],      vb[4 * VB_STRIDE]}      * bc(wt[4]) + acc;
        acc = v2f{va[5 * VB_STRIDE + 64], vb[5 * VB_STRIDE + 64]} * bc(wt[5]) + acc;
        acc = v2f{va[6 * VB_STRIDE],      vb[6 * VB_STRIDE]}      * bc(wt[6]) + acc;
        acc = v2f{va[7 * VB_STRIDE + 64], vb[7 * VB_STRIDE + 64]} * bc(wt[7]) + acc;
        acc = v2f{va[8 * VB_STRIDE],      vb[8 * VB_STRIDE]}      * bc(wt[8]) + acc;
        acc = v2f{va[9 * VB_STRIDE + 64], vb[9 * VB_STRIDE + 64]} * bc(wt[9]) + acc;
        if (scale_and_bias) acc = acc * bc(scale) + bc(bias);
        emit(i, n, acc.x);
        emit(i + 1, n, acc.y);
    }
    SSTAMP(4);
}

__device__ __forceinline__ void syn_hist_out(const SynWave &w, float *v_out, int lane)
{
    // new ring state: slots 31..23
#pragma unroll
    for (int r = 0; r < 18; r++)
        syn_st(v_out + lane + 64 * r, w.vb[(r >> 1) * VB_STRIDE + (r & 1) * 64 + lane]);
}

// The four pieces in sequence (stage-level kernel).
template <int UNROLL = 1, class SL, class Emit>
__device__ __forceinline__ void synth_channel(const SL &S, SynWave &w, const float *X0, const float *X1,
                                              const float *v_in, float *v_out,
                                              float scale, float bias, int lane, Emit emit)
{
    SSTAMP(0);
    SynIn d;
    syn_load<true>(X0, X1, v_in, lane, d);
    syn_rows<true>(S, w, d, lane);
    wave_sync();
    syn_poly<UNROLL>(S, w, scale, bias, lane, emit);
    syn_hist_out(w, v_out, lane);
    wave_sync();
    SSTAMP(5);
}

// k_synth: one wave per frame.  While a channel runs its polyphase sum, the X rows and ring state of
// the next unit -- the frame's other channel, or the first channel of the wave's next frame (the feed
// knows indices one frame ahead) -- are already on their way into registers.
template <int FMT>
__global__ __launch_bounds__(SYN_WAVES_F32 * WAVE)
void k_synth(const float *__restrict__ g_tab, const float *g_X,
             const float *g_state_in, float *g_state_out, int state_words, int off_syn0,
             int nout, void *__restrict__ g_pcm, float scale, float bias,
             unsigned long long n_frames, unsigned long long pcm_frame0, unsigned *g_queue,
             const unsigned char *__restrict__ g_xtop, const float *__restrict__ g_zero)
{
    constexpr int NW = SYN_WAVES_F32;
    __shared__ SynLdsT<NW> S;
    for (int i = threadIdx.x; i < 640; i += blockDim.x) S.win[i] = g_tab[TB_QMF_US + i];
    if (threadIdx.x < 64) S.rot[threadIdx.x] = g_tab[TB_ROT128S + threadIdx.x];
    if (threadIdx.x < 5) S.c16[threadIdx.x] = g_tab[TB_COS16 + threadIdx.x];
    if (threadIdx.x < 9) S.c32[threadIdx.x] = g_tab[TB_COS32 + threadIdx.x];
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE), lane = threadIdx.x % WAVE;
    SynWave &w = S.w[wave];
    // Bands the HF / PS stage did not store (they are +0: g_xtop, one byte per frame) come from a page of zeros: the
    // lanes that would load them aim at that page instead (an address select, no branch around the loads).
    auto load_unit = [&](unsigned long long f, int ch, SynIn &d) {
        const int xt = __builtin_amdgcn_readfirstlane((int)g_xtop[f * 2 + ch]);
        const float *X0 = (lane & 31) * 2 < xt ? g_X + (f * 2 + ch) * HE_X_CHANNEL : g_zero;
        syn_load<false>(X0, nullptr, g_state_in + f * state_words + off_syn0 + ch * HEAAC_ST_SYNTH, lane, d);
    };
    // the frame in work and the next one are known (FrameFeed, k_common.h: two frames per ticket)
    FrameFeed<2> feed;
    feed.init((unsigned long long)blockIdx.x * NW + wave, (unsigned long long)gridDim.x * NW, g_queue, lane);
    SynIn cur;
    if (feed.cur < n_frames) load_unit(feed.cur, 0, cur);
    while (feed.cur < n_frames) {
        const unsigned long long f = feed.cur, f1 = feed.nxt;
        feed.request(lane);
        float *st_out = g_state_out + f * state_words + off_syn0;
        // one channel: rows from `cur`, then the next unit's loads, then the polyphase sum
        auto channel = [&](int ch, auto emit) {
            SSTAMP(0);
            syn_rows<false>(S, w, cur, lane);
            wave_sync();
            if (ch + 1 < nout) load_unit(f, ch + 1, cur);
            else if (f1 < n_frames) load_unit(f1, 0, cur);
            syn_poly<1>(S, w, scale, bias, lane, emit);
            syn_hist_out(w, st_out + ch * HEAAC_ST_SYNTH, lane);
            wave_sync();
            SSTAMP(5);
        };
        if (FMT == HEAAC_PCM_F32_PLANAR) {
            for (int ch = 0; ch < nout; ch++) {
                float *o = reinterpret_cast<float *>(g_pcm) + ((pcm_frame0 + f) * nout + ch) * 2048;
                channel(ch, [&](int i, int n, float v) { syn_st(o + 64 * i + n, v); });
            }
        } else {
            // float_to_int16_interleave (dsputil.c:3989-4001) as 2-byte stores, one pass per channel: the left
            // samples of a frame go out first, the right ones fill the other halves of the same lines a few
            // microseconds later.  Most lines do NOT merge on the way: WRITE_SIZE shows 22.6 KiB per frame where 17.4
            // are algorithmic (profiles/traffic.json, hev2_s16; 25.4 before the loads above went non-temporal).  Packing (L, R) into one 4-byte store needs the left
            // channel's 4 KiB to wait somewhere, and there is nowhere: the v rows fill the LDS at seven waves (151 of
            // 160 KB; no row is dead while the right channel's sum starts), the registers are full at 256, and two
            // inlined instances of the polyphase sum with different store code cost 152 spilled VGPRs and 40 % of
            // this kernel's time.  So the channel loop keeps ONE instance of the sum, as the float path does.
            int16_t *o = reinterpret_cast<int16_t *>(g_pcm) + (pcm_frame0 + f) * 2048 * nout;
            for (int ch = 0; ch < nout; ch++)
                channel(ch, [&](int i, int n, float v) {
                    // (plain stores, not the non-temporal ones of the float path: -10 % kernel time for these 2-byte,
                    // half-line stores -- profiles/r04_experiments.md E3; the bytes written do not change)
                    o[(64 * i + n) * nout + ch] = (int16_t)pcm_int16<FMT>(v);
                });
        }
        feed.advance();
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

// Downsampled synthesis bank (div = 1, aacsbr.c:1175-1230): one wave per channel.  Lanes 0..31 run
// the slots' 128-point IMDCTs (one per slot), then every lane forms two output samples per pass:
// lane = (slot parity, n).
#define DS_STRIDE 65
#define DS_WAVES 8                // two waves per SIMD (212 - 256 VGPRs); 10.7 KB of LDS each
struct SynDsLds {
    float win[320];               // sbr_qmf_window_ds
    float rot[64], c16[8], c32[12];
    float vb[DS_WAVES][41 * DS_STRIDE];
};
__device__ __forceinline__ void syn_ds_lds_init(SynDsLds &S, const float *g_tab)
{
    for (int i = threadIdx.x; i < 320; i += blockDim.x) S.win[i] = g_tab[TB_QMF_DS + i];
    if (threadIdx.x < 64) S.rot[threadIdx.x] = g_tab[TB_ROT128S + threadIdx.x];
    if (threadIdx.x < 5) S.c16[threadIdx.x] = g_tab[TB_COS16 + threadIdx.x];
    if (threadIdx.x < 9) S.c32[threadIdx.x] = g_tab[TB_COS32 + threadIdx.x];
    __syncthreads();
}

// One channel of the downsampled bank.  X0 / X1: re / im planes, row stride 64 (bands 0..31 used);
// v_in / v_out: 576 floats; emit(i, n, value) receives out[32 i + n].
// PLANES = false: X0 is a channel of the hand-over workspace, [slot][band][re, im] (X1 unused).
template <bool PLANES, class Emit>
__device__ __forceinline__ void synth_ds_channel(const SynDsLds &S, float *vb, const float *X0, const float *X1,
                                                 const float *v_in, float *v_out, float scale, float bias,
                                                 int lane, Emit emit)
{
    // history: 9 slots of 64 behind the 32 new ones
    for (int t = lane; t < 576; t += WAVE) vb[(32 + (t >> 6)) * DS_STRIDE + (t & 63)] = v_in[t];
    if constexpr (PLANES) {
        if (lane < 32) {
            const int i = lane;
            const float *r0 = X0 + i * 64, *r1 = X1 + i * 64;
            float o[64];
            // X[0][i][n] = -X[0][i][n]; X[0][i][32+n] = X[1][i][31-n]
            imdct128_reg([&](int j) -> float { return j < 32 ? -r0[j] : r1[63 - j]; }, o, S.rot, S.c16, S.c32);
            float *v = vb + (31 - i) * DS_STRIDE;
#pragma unroll
            for (int k = 0; k < 32; k++) {
                v[k]      =  o[63 - 2 * k];
                v[63 - k] = -o[62 - 2 * k];
            }
        }
    } else {
        // The workspace rows [slot][band][re, im]: bands 0..31 of the 32 slots are 32 x 256 bytes.  The wave fetches
        // them together, 16 bytes per lane and load (a lane walking its own row pulls every line through the vector
        // L1 sixteen times; -4 % kernel time, r04_experiments.md E9), and hands each slot's row to the slot's lane
        // through the v rows' memory, which is free until then.
        {
            const f32x4 *src = reinterpret_cast<const f32x4 *>(X0);
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int piece = lane + 64 * q, slot = piece >> 4, part = piece & 15;
                const f32x4 t = src[slot * 32 + part];          // row stride 128 floats = 32 pieces; the first 16 hold bands 0..31
                float *d = vb + slot * DS_STRIDE + 4 * part;
                d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
            }
        }
        wave_sync();
        float x[64];
        if (lane < 32) {
            const float *row = vb + lane * DS_STRIDE;
#pragma unroll
            for (int j = 0; j < 64; j++) x[j] = row[j];
        }
        wave_sync();
        if (lane < 32) {
            const int i = lane;
            float o[64];
            // X[0][i][n] = -X[0][i][n]; X[0][i][32+n] = X[1][i][31-n]   (x[2 b] = re, x[2 b + 1] = im of band b)
            imdct128_reg([&](int j) -> float { return j < 32 ? -x[2 * j] : x[2 * (63 - j) + 1]; }, o, S.rot, S.c16, S.c32);
            float *v = vb + (31 - i) * DS_STRIDE;
#pragma unroll
            for (int k = 0; k < 32; k++) {
                v[k]      =  o[63 - 2 * k];
                v[63 - k] = -o[62 - 2 * k];
            }
        }
    }
    wave_sync();
    {
        const int nn = lane & 31, par = lane >> 5;
        float wt[10];
#pragma unroll
        for (int j = 0; j < 10; j++) wt[j] = S.win[32 * j + nn];
        const bool scale_and_bias = scale != 1.0f || bias != 0.0f;
        for (int i = par; i < 32; i += 2) {
            const float *v = vb + (31 - i) * DS_STRIDE + nn;
            float acc = v[0] * wt[0] + 0.0f;
#pragma unroll
            for (int j = 1; j < 10; j++) acc = v[j * DS_STRIDE + ((j & 1) ? 32 : 0)] * wt[j] + acc;
            if (scale_and_bias) acc = acc * scale + bias;
            emit(i, nn, acc);
        }
    }
    for (int t = lane; t < 576; t += WAVE) v_out[t] = vb[(t >> 6) * DS_STRIDE + (t & 63)];
    wave_sync();
}

__global__ __launch_bounds__(DS_WAVES * WAVE)
void k_qmf_synthesis_ds(const float *__restrict__ g_tab, const float *__restrict__ g_X /* [n][2][32][64] */,
                        const float *g_v_in, float *g_v_out, float *__restrict__ g_out,
                        float scale, float bias, unsigned long long n)
{
    __shared__ SynDsLds S;
    syn_ds_lds_init(S, g_tab);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE), lane = threadIdx.x % WAVE;
    for (unsigned long long u = (unsigned long long)blockIdx.x * DS_WAVES + wave; u < n;
         u += (unsigned long long)gridDim.x * DS_WAVES) {
        float *o = g_out + u * 1024;
        synth_ds_channel<true>(S, S.vb[wave], g_X + u * 4096, g_X + u * 4096 + 2048, g_v_in + u * 576, g_v_out + u * 576,
                         scale, bias, lane, [&](int i, int nn, float v) { o[32 * i + nn] = v; });
    }
}

// The downsampled bank inside the HE pipeline (ff_sbr_apply with ext_sample_rate < sbr->sample_rate,
// aacsbr.c:1719, 1194-1203): one wave per frame, X from the stage workspace, 1024 samples per channel.
// The ring state is the first 576 words of the channel's synthesis state; the rest passes through.
template <int FMT>
__global__ __launch_bounds__(DS_WAVES * WAVE)
void k_synth_ds(const float *__restrict__ g_tab, const float *g_X,
                const float *g_state_in, float *g_state_out, int state_words, int off_syn0,
                int nout, void *__restrict__ g_pcm, float scale, float bias, unsigned long long n_frames)
{
    __shared__ SynDsLds S;
    syn_ds_lds_init(S, g_tab);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE), lane = threadIdx.x % WAVE;
    for (unsigned long long f = (unsigned long long)blockIdx.x * DS_WAVES + wave; f < n_frames;
         f += (unsigned long long)gridDim.x * DS_WAVES) {
        for (int ch = 0; ch < nout; ch++) {
            const float *X0 = g_X + (f * 2 + ch) * HE_X_CHANNEL, *X1 = nullptr;
            const float *v_in = g_state_in + f * state_words + off_syn0 + ch * HEAAC_ST_SYNTH;
            float *v_out = g_state_out + f * state_words + off_syn0 + ch * HEAAC_ST_SYNTH;
            if (v_out != v_in)
                for (int t = 576 + lane; t < HEAAC_ST_SYNTH; t += WAVE) v_out[t] = v_in[t];
            if (FMT == HEAAC_PCM_F32_PLANAR) {
                float *o = reinterpret_cast<float *>(g_pcm) + (f * nout + ch) * 1024;
                synth_ds_channel<false>(S, S.vb[wave], X0, X1, v_in, v_out, scale, bias, lane,
                                 [&](int i, int nn, float v) { o[32 * i + nn] = v; });
            } else {
                int16_t *o = reinterpret_cast<int16_t *>(g_pcm) + f * 1024 * nout + ch;
                synth_ds_channel<false>(S, S.vb[wave], X0, X1, v_in, v_out, scale, bias, lane,
                                 [&](int i, int nn, float v) { o[(32 * i + nn) * nout] = (int16_t)pcm_int16<FMT>(v); });
            }
        }
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

// The product fuses the HF stage with baseline PS (k_hfps).  A -DHEAAC_TUNING build keeps the stages in
// separate kernels when HEAAC_HE_UNFUSED=1 is set (A/B measurements only).
static bool he_fused()
{
#ifdef HEAAC_TUNING
    static const bool fused = []() { const char *e = getenv("HEAAC_HE_UNFUSED"); return !(e && e[0] == '1'); }();
    return fused;
#else
    return true;
#endif
}

extern "C" int heaac_launch_he(const float *d_tab, const uint16_t *d_rev, int cfg,
                               const float *d_coeffs, const HeaacIcs *d_ics,
                               const HeaacSbrFrame *d_sbr, const HeaacSbrHeader *d_hdr, unsigned n_hdr,
                               const HeaacPsFrame *d_ps,
                               const float *d_state_in, float *d_state_out,
                               void *d_pcm, int pcm_format,
                               float *d_ws_W, float *d_ws_X, unsigned *d_queue,
                               unsigned char *d_xtop, const float *d_zero,
                               size_t n, size_t pcm_frame0, int flags, hipStream_t s)
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
    // ac->sf_scale / ac->add_bias: the C conversion's, or the SIMD configuration's (aacdec.c:573-581)
    const bool simd = pcm_format == HEAAC_PCM_S16_INTERLEAVED_SSE2;
    const float sf_scale = simd ? -1.0f / 1024.0f : HEAAC_SF_SCALE;

    // queue heads of the kernels that draw frames dynamically (k_hfps: [0], k_synth: [2]); the static
    // stride stays where it measured faster (k_core_ana, k_hfadj: neighbouring waves share lines)
    if (hipMemsetAsync(d_queue, 0, 64, s) != hipSuccess) return HEAAC_ERR_HIP;
    // every frame's X rows are whole (64 bands) unless the fused HF + PS kernel says otherwise
    if (hipMemsetAsync(d_xtop, 64, 2 * n, s) != hipSuccess) return HEAAC_ERR_HIP;
    hipLaunchKernelGGL(k_core_ana, dim3(he_grid((units + 1) / 2, CA_WAVES)), dim3(CA_WAVES * WAVE), 0, s,
                       d_tab, d_rev, d_coeffs, d_ics, d_state_in, d_state_out, words, ncore,
                       off_saved0, off_sbr0, d_ws_W, 1 / (-1024 * sf_scale), units);
    if (cfg == HEAAC_CFG_HEV2 && he_fused()) {
        // HF adjustment + baseline PS in one kernel; the general PS kernel finishes the
        // frames with another PS layout (it skips the rest)
        const int off_ps = off_syn0 + 2 * HEAAC_ST_SYNTH;
        int rc = heaac_launch_hfps(d_tab, d_sbr, d_hdr, n_hdr, d_ps, d_ws_W, d_state_in, d_state_out, words,
                                   off_sbr0, off_ps, d_ws_X, n, d_queue, d_xtop, s);
        if (rc != HEAAC_OK) return rc;
        rc = heaac_launch_ps(d_tab, d_ps, d_sbr, d_hdr, n_hdr, d_state_in, d_state_out, words, off_ps, d_ws_X, n, 2, s);
        if (rc != HEAAC_OK) return rc;
    } else {
        hipLaunchKernelGGL(k_hfadj, dim3(he_grid((units + 1) / 2, HF_WAVES)), dim3(HF_WAVES * WAVE), 0, s,
                           d_tab, d_sbr, d_hdr, n_hdr, d_ws_W, d_state_in, d_state_out, words, ncore, off_sbr0,
                           d_ws_X, units, d_queue + 1, cfg == HEAAC_CFG_HEV2 ? nullptr : d_xtop);
        if (cfg == HEAAC_CFG_HEV2) {
            int rc = heaac_launch_ps(d_tab, d_ps, d_sbr, d_hdr, n_hdr, d_state_in, d_state_out, words,
                                     off_syn0 + 2 * HEAAC_ST_SYNTH, d_ws_X, n, 3, s);
            if (rc != HEAAC_OK) return rc;
        }
    }
    const float scale = -1024 * sf_scale, bias = simd ? 0.0f : HEAAC_ADD_BIAS;
    if (flags & HEAAC_HE_DOWNSAMPLED) {
        const dim3 gd(he_grid(n, DS_WAVES)), bd(DS_WAVES * WAVE);
        char *pcm = (char *)d_pcm + pcm_frame0 * nout * 1024 * (pcm_format == HEAAC_PCM_F32_PLANAR ? 4 : 2);
        if (pcm_format == HEAAC_PCM_F32_PLANAR)
            hipLaunchKernelGGL((k_synth_ds<HEAAC_PCM_F32_PLANAR>), gd, bd, 0, s, d_tab, d_ws_X, d_state_in, d_state_out,
                               words, off_syn0, nout, (void *)pcm, scale, bias, (unsigned long long)n);
        else if (simd)
            hipLaunchKernelGGL((k_synth_ds<HEAAC_PCM_S16_INTERLEAVED_SSE2>), gd, bd, 0, s, d_tab, d_ws_X, d_state_in, d_state_out,
                               words, off_syn0, nout, (void *)pcm, scale, bias, (unsigned long long)n);
        else
            hipLaunchKernelGGL((k_synth_ds<HEAAC_PCM_S16_INTERLEAVED>), gd, bd, 0, s, d_tab, d_ws_X, d_state_in, d_state_out,
                               words, off_syn0, nout, (void *)pcm, scale, bias, (unsigned long long)n);
        return hipGetLastError() == hipSuccess ? HEAAC_OK : HEAAC_ERR_HIP;
    }
    const dim3 g(he_grid(n, SYN_WAVES_F32)), b(SYN_WAVES_F32 * WAVE);
    if (pcm_format == HEAAC_PCM_F32_PLANAR)
        hipLaunchKernelGGL((k_synth<HEAAC_PCM_F32_PLANAR>), g, b, 0, s, d_tab, d_ws_X, d_state_in, d_state_out,
                           words, off_syn0, nout, d_pcm, scale, bias,
                           (unsigned long long)n, (unsigned long long)pcm_frame0, d_queue + 2, d_xtop, d_zero);
    else if (pcm_format == HEAAC_PCM_S16_INTERLEAVED)
        hipLaunchKernelGGL((k_synth<HEAAC_PCM_S16_INTERLEAVED>), g, b, 0, s, d_tab, d_ws_X, d_state_in, d_state_out,
                           words, off_syn0, nout, d_pcm, scale, bias,
                           (unsigned long long)n, (unsigned long long)pcm_frame0, d_queue + 2, d_xtop, d_zero);
    else if (simd)
        hipLaunchKernelGGL((k_synth<HEAAC_PCM_S16_INTERLEAVED_SSE2>), g, b, 0, s, d_tab, d_ws_X, d_state_in, d_state_out,
                           words, off_syn0, nout, d_pcm, scale, bias,
                           (unsigned long long)n, (unsigned long long)pcm_frame0, d_queue + 2, d_xtop, d_zero);
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

extern "C" int heaac_launch_qmf_synthesis_ds(const float *d_tab, const float *d_X, const float *d_v_in,
                                            float *d_v_out, float *d_out, float scale, float bias,
                                            size_t n, hipStream_t s)
{
    if (n == 0) return HEAAC_OK;
    hipLaunchKernelGGL(k_qmf_synthesis_ds, dim3(he_grid(n, DS_WAVES)), dim3(DS_WAVES * WAVE), 0, s,
                       d_tab, d_X, d_v_in, d_v_out, d_out, scale, bias, (unsigned long long)n);
    return hipGetLastError() == hipSuccess ? HEAAC_OK : HEAAC_ERR_HIP;
}


#ifdef HEAAC_STAMPS
// accumulated phase timeline of this translation unit's kernels (k_synth): out[0..31] cycles, out[32] units
extern "C" int heaac_debug_timeline_he(unsigned long long *out)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tl_acc), sizeof(g_tl_acc)) != hipSuccess) return -1;
    return hipMemcpyFromSymbol(out + 32, HIP_SYMBOL(g_tl_cnt), sizeof(g_tl_cnt)) == hipSuccess ? 0 : -1;
}
#endif
