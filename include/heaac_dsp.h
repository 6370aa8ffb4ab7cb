/* heaac_dsp.h -- batched HE-AAC decode DSP, C ABI (MI355X / gfx950 build).
 *
 * This is the drop-in boundary for the ONE hot path this repo accelerates: the
 * post-parse float DSP of the reference AAC decoder,
 *   spectral_to_sample()            libavcodec/aacdec.c:1903-1933
 *     imdct_and_windowing()         libavcodec/aacdec.c:1741-1806
 *     ff_sbr_apply()                libavcodec/aacsbr.c:1716-1771
 *       ff_ps_apply()               libavcodec/aacps.c:973-992
 *   float_to_int16_interleave       libavcodec/dsputil.c:3972-4001
 *
 * The reference runs that path for ONE frame of ONE stream per call, on state
 * held inside AACContext/ChannelElement/SpectralBandReplication/PSContext.
 * Here every (state, frame) pair is an independent unit: state is an explicit
 * input and output record, so thousands of units run per kernel launch and a
 * batch shards across GPUs by index with no communication.
 *
 * Plain C: POD records, plain pointers, sizes, int status.  No torch types.
 * Pointers named d_* are DEVICE pointers (HBM); `stream` is a hipStream_t
 * passed as void* (NULL = default stream).  Nothing here allocates or
 * synchronises, so calls may be captured in a hipGraph.
 */
#ifndef HEAAC_DSP_H
#define HEAAC_DSP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ */
/* Constants of the path                                               */
/* ------------------------------------------------------------------ */

/* enum WindowSequence, libavcodec/aac.h:66-71 */
enum {
    HEAAC_ONLY_LONG_SEQUENCE   = 0,
    HEAAC_LONG_START_SEQUENCE  = 1,
    HEAAC_EIGHT_SHORT_SEQUENCE = 2,
    HEAAC_LONG_STOP_SEQUENCE   = 3,
};

/* The reference's pure-C float path (aacdec.c:573-576): samples carry a
 * +385.0 bias and a 2^-15 scale so that float_to_int16_one() can read the
 * int16 straight out of the mantissa. */
#define HEAAC_ADD_BIAS   385.0f
#define HEAAC_SF_SCALE   (-1.0f / (1024.0f * 32768.0f))   /* ac->sf_scale */

/* Workload kinds (BASELINE.json configs) */
enum {
    HEAAC_CFG_LC_MONO   = 0,  /* SCE, AAC-LC                    -> 1 x 1024 */
    HEAAC_CFG_LC_STEREO = 1,  /* CPE, AAC-LC                    -> 2 x 1024 */
    HEAAC_CFG_HEV1      = 2,  /* CPE + SBR x2                   -> 2 x 2048 */
    HEAAC_CFG_HEV2      = 3,  /* SCE + SBR + Parametric Stereo  -> 2 x 2048 */
    HEAAC_CFG_HEV1_MONO = 4,  /* SCE + SBR                      -> 1 x 2048 */
};

/* PCM output formats */
enum {
    HEAAC_PCM_F32_PLANAR = 0, /* the reference's `ret[]` floats (bias included),
                                 [frame][ch][len]                          */
    HEAAC_PCM_S16_INTERLEAVED = 1, /* avcodec_decode_audio3 output:
                                 [frame][len][ch] int16 (dsputil.c:3989)   */
    /* The decoder as it is configured when float_to_int16_interleave is one of the x86 SIMD versions
     * (aacdec.c:577-581: add_bias = 0, sf_scale = 1 / -1024, sf_offset = 60 -- the CALLER's spectrum is
     * 32768 x the C path's, an exact power of two) with float_to_int16_interleave_sse2's conversion
     * (x86/dsputil_mmx.c:2356-2372, 2405-2436): cvtps2dq = round to nearest even, NaN and |x| >= 2^31 give
     * 0x80000000; packssdw saturates.  Same layout as HEAAC_PCM_S16_INTERLEAVED.  (AAC-Main prediction reads
     * sf_scale too, aacdec.c:1285-1287: heaac_spectral_tools_batch computes it for the C path only.) */
    HEAAC_PCM_S16_INTERLEAVED_SSE2 = 2,
};

/* ------------------------------------------------------------------ */
/* P: per-frame parameters produced by the (host) parsers              */
/* ------------------------------------------------------------------ */

/* IndividualChannelStream fields the DSP reads (aac.h:137-138). */
typedef struct HeaacIcs {
    uint8_t window_sequence[2];   /* [0] this frame, [1] previous frame */
    uint8_t use_kb_window[2];     /* [0] this frame, [1] previous frame */
} HeaacIcs;                       /* 4 bytes */

/* Header-derived SBR tables (sbr.h:122-156), one record per distinct SBR
 * header; frames refer to it by index.  All band borders are <= 64. */
typedef struct HeaacSbrHeader {
    uint8_t k0;                   /* sbr->k[0]                              */
    uint8_t k2;                   /* sbr->k[2]                              */
    uint8_t kx;                   /* sbr->kx[1]                             */
    uint8_t m;                    /* sbr->m[1]                              */
    uint8_t n[2];                 /* N_low, N_high                          */
    uint8_t n_q;                  /* noise floor bands  (<= 5)              */
    uint8_t n_lim;                /* limiter bands      (<= 28)             */
    uint8_t n_master;
    uint8_t num_patches;          /* <= 6                                   */
    uint8_t bs_limiter_gains;     /* 0..3                                   */
    uint8_t bs_interpol_freq;
    uint8_t bs_smoothing_mode;
    uint8_t bs_amp_res_header;
    uint8_t pad0[2];
    uint8_t patch_num_subbands[6];
    uint8_t patch_start_subband[6];
    uint8_t f_tablenoise[6];
    uint8_t pad1[2];
    uint8_t f_tablelow[28];       /* 25 used */
    uint8_t f_tablehigh[52];      /* 49 used */
    uint8_t f_tablelim[32];       /* 29 used */
    /* Per-QMF-band lookups derived from the tables above by heaac_sbr_make_header()
     * (0xff = none).  They replace the reference's per-frame table searches in
     * sbr_hf_gen / sbr_mapping / sbr_gain_calc (aacsbr.c:1366-1376, 1457-1492, 1563). */
    uint8_t map_hi[64];           /* i: f_tablehigh[i] <= k < f_tablehigh[i+1]            */
    uint8_t map_lo[64];           /* i: f_tablelow[i]  <= k < f_tablelow[i+1]             */
    uint8_t map_nq[64];           /* i: f_tablenoise[i] <= k < f_tablenoise[i+1]          */
    uint8_t map_lim[64];          /* i: f_tablelim[i]  <= k < f_tablelim[i+1]             */
    uint8_t map_mid[64];          /* i: k == (f_tablehigh[i] + f_tablehigh[i+1]) >> 1     */
    uint8_t map_src[64];          /* patch source band p of HF band k (sbr_hf_gen)        */
} HeaacSbrHeader;                 /* 532 bytes */

/* Per-channel SBR frame data (SBRData bitstream fields, sbr.h:64-73,84,99-105)
 * exactly as read_sbr_grid()/read_sbr_envelope()/read_sbr_noise() leave them
 * BEFORE sbr_dequant(): env/noise scalefactors are still the accumulated
 * integers. */
typedef struct HeaacSbrChannel {
    uint8_t bs_num_env;           /* 1..5                                   */
    uint8_t bs_num_noise;         /* 1..2                                   */
    uint8_t bs_amp_res;
    uint8_t bs_add_harmonic_flag;
    uint8_t bs_freq_res[8];       /* [1..bs_num_env]; [0] = previous frame  */
    uint8_t t_env[8];             /* [0..bs_num_env]                        */
    uint8_t t_q[3];
    uint8_t t_env_num_env_old;    /* t_env[bs_num_env] of the previous frame*/
    int8_t  e_a[2];               /* l_APrev, l_A                           */
    uint8_t bs_invf_mode[2][5];   /* [0] this frame, [1] previous frame     */
    uint8_t bs_add_harmonic[48];
    uint8_t env_facs_q[5][48];    /* env_facs[1..5][k] before dequant       */
    uint8_t noise_facs_q[2][5];   /* noise_facs[1..2][k] before dequant     */
    uint8_t pad[2];
} HeaacSbrChannel;                /* 336 bytes */

typedef struct HeaacSbrFrame {
    uint16_t hdr;                 /* index into the HeaacSbrHeader table    */
    uint8_t  start;               /* sbr->start                             */
    uint8_t  reset;               /* sbr->reset (header changed this frame) */
    uint8_t  kx_old;              /* sbr->kx[0]                             */
    uint8_t  m_old;               /* sbr->m[0]                              */
    uint8_t  bs_coupling;
    uint8_t  pad;
    HeaacSbrChannel ch[2];
} HeaacSbrFrame;                  /* 680 bytes */

/* PSContext bitstream fields (aacps.h:41-61) as ff_ps_read_data() leaves them. */
typedef struct HeaacPsFrame {
    uint8_t start;                /* ps->start; 0 = copy mono to both       */
    uint8_t is34bands;
    uint8_t is34bands_old;
    uint8_t num_env;              /* 1..5 after the envelope fix-up         */
    uint8_t num_env_old;
    uint8_t enable_ipdopd;
    uint8_t iid_quant;
    uint8_t icc_mode;
    uint8_t nr_iid_par;           /* 10, 20, 34                             */
    uint8_t nr_icc_par;
    uint8_t nr_ipdopd_par;        /* 5, 11, 17                              */
    uint8_t pad;
    int8_t  border_position[8];   /* [0..num_env]; [0] = -1                 */
    int8_t  iid_par[5][34];
    int8_t  icc_par[5][34];
    int8_t  ipd_par[5][17];
    int8_t  opd_par[5][17];
    uint8_t pad2[2];
} HeaacPsFrame;                   /* 532 bytes */

/* ------------------------------------------------------------------ */
/* S: inter-frame state records (float32 words; ints stored bit-exact)  */
/* ------------------------------------------------------------------ */
/* Every record is the minimal live state, not the reference's padded arrays.
 * Offsets are in 32-bit words inside one frame's record.                  */

/* --- one AAC core channel: sce->saved[0..511] (aac.h:214) --- */
#define HEAAC_ST_SAVED            512

/* --- one SBR channel (SBRData state, sbr.h:80-105) --- */
#define HEAAC_SBR_XHIST           0      /* 288: analysis_filterbank_samples tail   */
#define HEAAC_SBR_WTAIL           288    /* 8*32*2: W[1][24..31][k][re,im]          */
#define HEAAC_SBR_YTAIL           800    /* 6*64*2: Y[1][32..37][k][re,im]          */
#define HEAAC_SBR_GTAIL           1568   /* 4*48: g_temp rows 2*t_env[L]+0..3       */
#define HEAAC_SBR_QTAIL           1760   /* 4*48: q_temp rows                       */
#define HEAAC_SBR_BW              1952   /* 5: bw_array                             */
#define HEAAC_SBR_IDXNOISE        1957   /* 1 (u32): f_indexnoise                   */
#define HEAAC_SBR_IDXSINE         1958   /* 1 (u32): f_indexsine                    */
#define HEAAC_SBR_SIDX            1959   /* 12 words = 48 bytes: s_indexmapped[0]   */
#define HEAAC_SBR_PAD             1971
#define HEAAC_ST_SBR              1972

/* --- one synthesis filterbank (output channel): the 9 most recent v slots,
 *     newest first = synthesis_filterbank_samples[v_off .. v_off+1151] --- */
#define HEAAC_ST_SYNTH            1152

/* --- Parametric Stereo (PSContext state, aacps.h:63-74) --- */
#define HEAAC_PS_INBUF            0      /* 5*6*2: in_buf[i][0..5]                  */
#define HEAAC_PS_DELAY            60     /* 14*91*2: delay[k][32+j] stored [j][k][re,im] (band-fastest) */
#define HEAAC_PS_APDELAY          2608   /* 3*5*50*2: ap_delay[k][m][32+j] stored [m][j][k][re,im]     */
#define HEAAC_PS_PEAK             4108   /* 34: peak_decay_nrg                      */
#define HEAAC_PS_PSMOOTH          4142   /* 34: power_smooth                        */
#define HEAAC_PS_PDIFF            4176   /* 34: peak_decay_diff_smooth              */
#define HEAAC_PS_H                4210   /* 4*2*34: H11,H12,H21,H22 [re/im][b] of the last envelope */
#define HEAAC_PS_HIST             4482   /* 18 words: opd_hist[34], ipd_hist[34] (int8) + pad */
#define HEAAC_ST_PS               4500

/* Per-frame state record sizes (32-bit words) by workload. */
#define HEAAC_STATE_WORDS_LC_MONO   (HEAAC_ST_SAVED)
#define HEAAC_STATE_WORDS_LC_STEREO (2 * HEAAC_ST_SAVED)
/* HEv1 stereo: [saved0|saved1|sbr0|sbr1|synth0|synth1] */
#define HEAAC_STATE_WORDS_HEV1      (2 * HEAAC_ST_SAVED + 2 * HEAAC_ST_SBR + 2 * HEAAC_ST_SYNTH)
/* HEv1 mono:   [saved0|sbr0|synth0] */
#define HEAAC_STATE_WORDS_HEV1_MONO (HEAAC_ST_SAVED + HEAAC_ST_SBR + HEAAC_ST_SYNTH)
/* HEv2:        [saved0|sbr0|synthL|synthR|ps] */
#define HEAAC_STATE_WORDS_HEV2      (HEAAC_ST_SAVED + HEAAC_ST_SBR + 2 * HEAAC_ST_SYNTH + HEAAC_ST_PS)

/* ------------------------------------------------------------------ */
/* Batched entry points                                                */
/* ------------------------------------------------------------------ */

/* Opaque device-side context: immutable tables (twiddles, windows, QMF
 * prototype, noise table, PS filters) resident in HBM for one device, plus a
 * scratch workspace sized for `max_frames`.  One per GPU/process.  The HE
 * calls pass intermediates (and a frame queue) through that workspace, so the
 * HE calls on one HeaacDevice must not overlap: the context remembers the stream
 * of its last HE call, and an HE call on ANOTHER stream while that work is still
 * in flight returns HEAAC_ERR_ARG (it is accepted once the earlier work has
 * completed; inside a stream capture the question cannot be asked, so a capture
 * must run on the stream of the context's previous HE call -- warm up on the
 * capture stream).  Concurrent streams take one HeaacDevice each. */
typedef struct HeaacDevice HeaacDevice;

/* Create the context on the current HIP device.  Returns 0 on success,
 * a negative HEAAC_ERR_* otherwise.  Replaces the table set-up done by
 * aac_decode_init() (aacdec.c:583-598), ff_aac_sbr_init() (aacsbr.c:86-126),
 * ff_aac_sbr_ctx_init() (aacsbr.c:128-137) and ps_tableinit()
 * (aacps_tablegen.h:80-209). */
int heaac_device_create(HeaacDevice **out, size_t max_frames);
void heaac_device_destroy(HeaacDevice *dev);

/* Bytes of device workspace heaac_device_create() allocates for max_frames. */
size_t heaac_device_workspace_bytes(size_t max_frames);

enum {
    HEAAC_OK            =  0,
    HEAAC_ERR_ARG       = -1,   /* bad argument / unsupported configuration */
    HEAAC_ERR_HIP       = -2,   /* a HIP runtime call failed               */
    HEAAC_ERR_NOMEM     = -3,
    HEAAC_ERR_NODEVICE  = -4,   /* no gfx950 device visible                */
};
const char *heaac_strerror(int err);

/* Batched ff_imdct_half (mdct.c:124-159): n independent transforms.
 * which: 0 = (11,1,1.0)  AAC long      aacdec.c:590
 *        1 = (8,1,1.0)   AAC short     aacdec.c:591
 *        2 = (7,1,1/64)  SBR synthesis aacsbr.c:134
 *        3 = (7,1,-2.0)  SBR analysis  aacsbr.c:135
 * d_in / d_out: [n][N/2] floats, must not alias. */
int heaac_imdct_half_batch(HeaacDevice *dev, int which,
                           float *d_out, const float *d_in,
                           size_t n, void *stream);

/* Batched AAC-LC channel-element synthesis = imdct_and_windowing() for every
 * channel (+ float_to_int16_interleave when pcm_format is S16).
 *   d_coeffs   [n][channels][1024]  sce->coeffs after the spectral tools
 *   d_ics      [n][channels]        HeaacIcs
 *   d_state_in/out [n][channels*512] saved[]  (may alias each other)
 *   d_pcm      F32: [n][channels][1024] floats (bias 385 included)
 *              S16: [n][1024][channels] int16
 */
int heaac_lc_decode_batch(HeaacDevice *dev, int channels,
                          const float *d_coeffs, const HeaacIcs *d_ics,
                          const float *d_state_in, float *d_state_out,
                          void *d_pcm, int pcm_format,
                          size_t n, void *stream);

/* AFTER_IMDCT independent channel coupling: apply_independent_coupling()
 * (aacdec.c:1849-1862) the way spectral_to_sample() applies it to an SCE / CPE once the element's own
 * IMDCT is done (:1929-1930, apply_channel_coupling :1870-1898), AAC-LC (no SBR: len = 1024):
 *     dest[i] += gain * (src[i] - bias)          bias = HEAAC_ADD_BIAS
 *   d_pcm      [n][channels][1024]  target->ret: heaac_lc_decode_batch's F32 output, updated in place
 *   d_cce      [n][1024]            the coupling element's ret: heaac_lc_decode_batch, channels = 1, F32
 *   d_coupling [n]                  per target channel: coupled or not, cce->coup.gain[index][0]
 *   d_s16      NULL, or [n][1024][channels] int16: float_to_int16_interleave of the result
 * Several coupling elements on one target = several calls, in element order. */
typedef struct HeaacCoupling {
    float   gain[2];
    uint8_t on[2];
    uint8_t pad[2];
} HeaacCoupling;                  /* 12 bytes */

/* ------------------------------------------------------------------------
 * float_to_int16_interleave for any channel count (dsputil.c:3989-4001 as aac_decode_frame calls it, aacdec.c:2096-2097:
 * `output_data[]` = one plane per output channel in layout order).  The planes are the F32 outputs of the decode
 * calls of a layout's elements (heaac_lc_decode_batch / heaac_he_decode_batch with HEAAC_PCM_F32_PLANAR: bias
 * included): channel c of frame f starts at planes[c].d_base + f * planes[c].frame_stride and has `len` floats
 * (1024, or 2048 behind SBR; a multiple of 4, bases and strides 16-byte aligned).
 *   d_out [n][len][channels] int16;  pcm_format: HEAAC_PCM_S16_INTERLEAVED or HEAAC_PCM_S16_INTERLEAVED_SSE2. */
#define HEAAC_MAX_PCM_PLANES 16
typedef struct HeaacPlaneRef {
    const float *d_base;
    size_t frame_stride;          /* floats between the same channel of consecutive frames */
} HeaacPlaneRef;
int heaac_pcm_interleave_batch(HeaacDevice *dev, int channels, const HeaacPlaneRef *planes, int len,
                               int pcm_format, int16_t *d_out, size_t n, void *stream);

int heaac_couple_after_imdct_batch(HeaacDevice *dev, int channels, float *d_pcm, const float *d_cce,
                                   const HeaacCoupling *d_coupling, int16_t *d_s16,
                                   size_t n, void *stream);

/* Batched HE-AAC channel-element synthesis = imdct_and_windowing() (bias 0)
 * + ff_sbr_apply() (+ ff_ps_apply()) (+ float_to_int16_interleave).
 *   cfg        HEAAC_CFG_HEV1 (CPE), HEAAC_CFG_HEV1_MONO (SCE) or HEAAC_CFG_HEV2 (SCE+PS)
 *   d_coeffs   [n][core_channels][1024]
 *   d_ics      [n][core_channels]
 *   d_sbr      [n] HeaacSbrFrame ; d_hdr: header table (n_hdr records)
 *   d_ps       [n] HeaacPsFrame (HEV2 only, else NULL)
 *   d_state_in/out [n][HEAAC_STATE_WORDS_*] (may alias each other)
 *   d_pcm      F32: [n][out_channels][2048] ; S16: [n][2048][out_channels]
 */
int heaac_he_decode_batch(HeaacDevice *dev, int cfg,
                          const float *d_coeffs, const HeaacIcs *d_ics,
                          const HeaacSbrFrame *d_sbr,
                          const HeaacSbrHeader *d_hdr, size_t n_hdr,
                          const HeaacPsFrame *d_ps,
                          const float *d_state_in, float *d_state_out,
                          void *d_pcm, int pcm_format,
                          size_t n, void *stream);

/* The same with options.  flags:
 *   HEAAC_HE_DOWNSAMPLED  the output runs at the CORE rate: ff_sbr_apply selects the 32-band synthesis
 *                         bank (div = 1, aacsbr.c:1719, 1194-1203) when m4ac.ext_sample_rate <
 *                         sbr->sample_rate.  d_pcm then holds 1024 samples per channel
 *                         (F32: [n][out_channels][1024], S16: [n][1024][out_channels]); of each channel's
 *                         synthesis state the first 576 words are the ring, the rest passes through. */
enum { HEAAC_HE_DOWNSAMPLED = 1 };
int heaac_he_decode_batch_ex(HeaacDevice *dev, int cfg, int flags,
                             const float *d_coeffs, const HeaacIcs *d_ics,
                             const HeaacSbrFrame *d_sbr,
                             const HeaacSbrHeader *d_hdr, size_t n_hdr,
                             const HeaacPsFrame *d_ps,
                             const float *d_state_in, float *d_state_out,
                             void *d_pcm, int pcm_format,
                             size_t n, void *stream);

/* Record validation.  The batched entry points take records from ANY parser (the library's own, heaac_parse.h,
 * only writes records that pass), so the per-frame records are where malformed data would arrive; the rules are the reference parser's own rejections
 * (read_sbr_grid aacsbr.c:609-745, sbr_make_f_master / sbr_make_f_derived :296-593, ff_ps_read_data
 * aacps.c:150-279) plus the bounds the kernels index with.  heaac_he_decode_batch does NOT validate:
 * it clamps what forms a global address (a bad record gives wrong audio for that frame, never a
 * fault), and expects untrusted input to have passed one of these first. */
/* first rule a record breaks (0 = valid) */
enum {
    HEAAC_BAD_NONE = 0,
    HEAAC_BAD_HDR_INDEX,        /* frame.hdr >= n_hdr                                        */
    HEAAC_BAD_HDR_RANGE,        /* kx > 32, m > 48, kx + m > 64, k0 > 32 (aacsbr.c:499-507)   */
    HEAAC_BAD_HDR_COUNTS,       /* n[0] > 24, n[1] > 48, n_q > 5, n_lim > 29, patches > 5     */
    HEAAC_BAD_HDR_TABLE,        /* a frequency table is not increasing from kx to kx + m      */
    HEAAC_BAD_HDR_MAP,          /* a per-band lookup points outside its table                 */
    HEAAC_BAD_HDR_FLAGS,
    HEAAC_BAD_HDR_UNSTARTED,    /* start = 1 on a header without SBR range (m = 0)            */
    HEAAC_BAD_SBR_NUM_ENV,      /* bs_num_env not in 1..5, bs_num_noise not in 1..2 (:627, :684) */
    HEAAC_BAD_SBR_T_ENV,        /* borders not increasing, t_env[L] > 19, or > 16 + 3 slots   */
    HEAAC_BAD_SBR_T_Q,          /* noise borders not borders of the envelope grid             */
    HEAAC_BAD_SBR_FLAGS,        /* freq_res / amp_res / invf_mode / e_a / coupling out of range */
    HEAAC_BAD_SBR_OLD_RANGE,    /* kx_old > 32 or kx_old + m_old > 64, t_env_num_env_old > 19 */
    HEAAC_BAD_PS_NUM_ENV,       /* num_env not in 1..5 (0 only as num_env_old)                */
    HEAAC_BAD_PS_BORDER,        /* border[0] != -1, not increasing, last != 31                */
    HEAAC_BAD_PS_NR_PAR,        /* nr_*_par outside {10, 20, 34} / {5, 11, 17}, modes > 5     */
    HEAAC_BAD_PS_PAR,           /* |iid| > 7 + 8 quant, icc not in 0..7, ipd / opd not in 0..7 */
};


/* One frame's records on the host: 0 if valid, else the HEAAC_BAD_* rule broken first (-1: bad call). */
int heaac_validate_frame(int cfg, const HeaacSbrFrame *sbr, const HeaacSbrHeader *hdr, size_t n_hdr,
                         const HeaacPsFrame *ps);

/* A batch of device-resident records (the arguments of heaac_he_decode_batch).  Returns HEAAC_OK, or
 * HEAAC_ERR_ARG with *first_bad = lowest failing frame index and *rule = its HEAAC_BAD_* code (either
 * pointer may be NULL).  Synchronises `stream`. */
int heaac_he_check_batch(HeaacDevice *dev, int cfg, const HeaacSbrFrame *d_sbr,
                         const HeaacSbrHeader *d_hdr, size_t n_hdr, const HeaacPsFrame *d_ps,
                         size_t n, void *stream, size_t *first_bad, int *rule);

/* Stage-level entry points (same kernels, exposed for parity tests and for
 * hosts that keep part of the pipeline themselves). */

/* sbr_qmf_analysis (aacsbr.c:1136-1169) for n channels:
 *   d_in [n][1024] core samples (bias 0), d_xhist_in/out [n][288],
 *   d_W [n][32][32][2]; scale = 1/(-1024*sf_scale) = 32768 on the C path. */
int heaac_qmf_analysis_batch(HeaacDevice *dev, const float *d_in,
                             const float *d_xhist_in, float *d_xhist_out,
                             float *d_W, float scale, size_t n, void *stream);

/* sbr_qmf_synthesis (aacsbr.c:1175-1230), div = 0, for n channels:
 *   d_X [n][2][32][64] (re plane, im plane; slots 0..31 of X[2][38][64]),
 *   d_v_in/out [n][1152], d_out [n][2048]; out = acc*scale + bias. */
int heaac_qmf_synthesis_batch(HeaacDevice *dev, const float *d_X,
                              const float *d_v_in, float *d_v_out,
                              float *d_out, float scale, float bias,
                              size_t n, void *stream);

/* Downsampled synthesis bank: sbr_qmf_synthesis with div = 1 (aacsbr.c:1175-1230), what
 * ff_sbr_apply selects when the output runs at the core rate (aacsbr.c:1719):
 *   d_X [n][2][32][64] as above (bands 0..31 of each row are used),
 *   d_v_in/out [n][576], d_out [n][1024]; out = acc*scale + bias.
 * The fused heaac_he_decode_batch implements the dual-rate bank (div = 0). */
int heaac_qmf_synthesis_ds_batch(HeaacDevice *dev, const float *d_X,
                                 const float *d_v_in, float *d_v_out,
                                 float *d_out, float scale, float bias,
                                 size_t n, void *stream);

/* ------------------------------------------------------------------------
 * Spectral tools that run on the dequantised spectrum before the IMDCT
 * (SURVEY s8f N1): apply_mid_side_stereo (aacdec.c:1390-1411),
 * apply_intensity_stereo (:1420-1451) and apply_tns (:1698-1736, with
 * compute_lpc_coefs, lpc.h:61-103).  With them the host/GPU boundary moves up
 * to "dequantised spectrum + side info".
 * ------------------------------------------------------------------------ */
enum { HEAAC_NOISE_BT = 13, HEAAC_INTENSITY_BT2 = 14, HEAAC_INTENSITY_BT = 15 };   /* aac.h:73-80 */
#define HEAAC_TNS_MAX_ORDER 20

/* the fields of IndividualChannelStream the tools read (aac.h:128-152) */
typedef struct HeaacToolsIcs {
    uint8_t  num_windows;             /* 1 or 8 */
    uint8_t  num_window_groups;
    uint8_t  max_sfb;
    uint8_t  num_swb;
    uint8_t  tns_max_bands;
    uint8_t  pad[3];
    uint8_t  group_len[8];
    uint16_t swb_offset[64];          /* ics->swb_offset[0 .. num_swb] */
} HeaacToolsIcs;                      /* 144 B */

/* TemporalNoiseShaping (aac.h:178-185) */
typedef struct HeaacTns {
    uint8_t present;
    uint8_t n_filt[8];
    uint8_t length[8][4];
    uint8_t direction[8][4];
    uint8_t order[8][4];
    uint8_t pad[3];
    float   coef[8][4][HEAAC_TNS_MAX_ORDER];
} HeaacTns;                           /* 2668 B */

/* AAC-Main backward-adaptive prediction side info (aac.h:146-149) */
typedef struct HeaacPrediction {
    uint8_t predictor_present;
    uint8_t predictor_reset_group;    /* 0: none, else 1..30 */
    uint8_t pred_sfb_max;             /* ff_aac_pred_sfb_max[sampling_index] (aactab.c:47-49) */
    uint8_t pad;
    uint8_t prediction_used[44];      /* [41] in the reference */
} HeaacPrediction;                    /* 48 B */

/* PredictorState (aac.h:114-122): one per spectral line below the prediction limit */
#define HEAAC_MAX_PREDICTORS 672
typedef struct HeaacPredictorState { float cor0, cor1, var0, var1, r0, r1; } HeaacPredictorState;

typedef struct HeaacToolsChannel {
    HeaacToolsIcs ics;
    uint8_t  band_type[128];          /* sce->band_type[idx], idx = g * max_sfb + sfb.  The runs the
                                         reference walks (band_type_run_end) are implied: all bands of a
                                         run share one type, so testing every band is the same walk. */
    float    sf[128];                 /* sce->sf[idx] (first 120 used) */
    HeaacTns tns;
    HeaacPrediction pred;
} HeaacToolsChannel;                  /* 3500 B */

typedef struct HeaacToolsFrame {
    uint8_t  common_window;           /* CPE only: both channels share ch[0].ics */
    uint8_t  ms_present;              /* 0: off, 1: ms_mask, 2: all bands (mask already all ones) */
    uint8_t  pad[2];
    uint8_t  ms_mask[128];            /* cpe->ms_mask[idx] */
    HeaacToolsChannel ch[2];
} HeaacToolsFrame;                    /* 7132 B */

/* In place on d_coeffs [n][channels][1024].  channels == 2: M/S (if common_window and
 * ms_present), intensity stereo, then TNS per channel -- the order of decode_cpe
 * (aacdec.c:1480-1492) followed by spectral_to_sample (:1913-1916).  channels == 1: TNS of
 * ch[0] only.
 * Perceptual noise substitution (the NOISE_BT branch of decode_spectrum_and_dequant,
 * aacdec.c:1016-1029) runs first when d_rng_in is not NULL: every band of type NOISE_BT is
 * filled from the stream's generator (lcg_random, aacdec.c:502-505; ac->random_state starts at
 * 0x1f2e3d4c, :567) in parse order -- channel 0 then channel 1, bands in (group, sfb) order, the
 * windows of a group, ascending k -- and scaled to sf[idx] / sqrtf(energy).  d_rng_in[n] is the
 * state before the frame, d_rng_out[n] the state after it (may alias).  With d_rng_in == NULL
 * noise bands are taken as given.
 * AAC-Main prediction (apply_prediction / predict, aacdec.c:1247-1322) runs when d_pred_in is not
 * NULL: d_pred_in/out [n][channels][672] predictor states (may alias; a stream starts from
 * reset_all_predictors: cor = r = 0, var = 1, aacdec.c:507-522).  Order as in the reference:
 * after noise substitution for channels without common_window (decode_ics, :1381-1382), after
 * M/S and before intensity stereo for a common-window pair (decode_cpe, :1483-1489). */
int heaac_spectral_tools_batch(HeaacDevice *dev, int channels, float *d_coeffs,
                               const HeaacToolsFrame *d_tools,
                               const int32_t *d_rng_in, int32_t *d_rng_out,
                               const HeaacPredictorState *d_pred_in, HeaacPredictorState *d_pred_out,
                               size_t n, void *stream);

/* ------------------------------------------------------------------------
 * Dependent channel coupling: a coupling_channel_element whose spectrum is added into its target
 * channels BEFORE the IMDCT -- apply_dependent_coupling (aacdec.c:1813-1843) under apply_channel_coupling
 * (:1870-1898), at the two points spectral_to_sample knows (:1912, :1917): before the target's TNS and between TNS
 * and IMDCT.  (The third point, AFTER_IMDCT, is heaac_couple_after_imdct_batch above.)
 * ------------------------------------------------------------------------ */
enum { HEAAC_CC_BEFORE_TNS = 0, HEAAC_CC_BETWEEN_TNS_AND_IMDCT = 1, HEAAC_CC_AFTER_IMDCT = 3 };   /* aac.h:83-87 */
#define HEAAC_MAX_CCE 16              /* coupling elements per access unit carried by the batched records */
#define HEAAC_MAX_CCE_LINKS 4         /* gain lists of one coupling element that land on the (one) target element */

/* One gain list of a coupling element applied to one channel of the target element: coup->gain[index][] with the
 * list index resolved against the target as apply_channel_coupling resolves it. */
typedef struct HeaacCceLink {
    uint8_t target_ch;                /* channel of the target element: 0 or 1 */
    uint8_t pad[3];
    float   gain[120];                /* [idx], idx = g * max_sfb + sfb of the COUPLING channel's grouping;
                                         AFTER_IMDCT: gain[0] only (decode_cce, aacdec.c:1541-1543) */
} HeaacCceLink;                       /* 484 B */

typedef struct HeaacCceFrame {
    uint8_t present;                  /* 0: this slot holds no coupling element in this access unit */
    uint8_t elem_id;                  /* instance tag; slots are filled in ascending tag order, the order in which
                                         apply_channel_coupling walks ac->che[TYPE_CCE][] */
    uint8_t coupling_point;           /* HEAAC_CC_* */
    uint8_t n_links;
    uint8_t behind_target;            /* 1: the element follows the target element in the access unit (order of the
                                         noise generator across elements) */
    uint8_t seq;                      /* position among the access unit's coupling elements in BITSTREAM order (the
                                         slots are in tag order) */
    uint8_t outputs_before;           /* output elements (SCE / CPE / LFE) in front of it in the access unit: with
                                         HeaacAacElementInfo.seq and `seq` the order of ALL elements, which is the
                                         order the noise generator runs through them */
    uint8_t pad;
    HeaacToolsIcs ics;                /* the coupling channel's own grouping and band offsets */
    uint8_t band_type[128];           /* its band types: ZERO_BT (0) bands couple nothing (:1828) */
    HeaacCceLink link[HEAAC_MAX_CCE_LINKS];
} HeaacCceFrame;                      /* 2216 B */

/* heaac_spectral_tools_batch in two halves, with dependent coupling in the second:
 *   HEAAC_TOOLS_PRE   noise substitution, AAC-Main prediction, M/S, intensity stereo -- what the reference does
 *                     while it parses an element (decode_ics / decode_cpe)
 *   HEAAC_TOOLS_POST  what spectral_to_sample does to the element before its IMDCT (aacdec.c:1911-1918): coupling
 *                     at BEFORE_TNS, TNS, coupling at BETWEEN_TNS_AND_IMDCT
 * d_cce [n][n_cce] / d_cce_coeffs [n][n_cce][1024]: the access unit's coupling elements (slots in ascending tag
 * order) and their spectra AFTER their own tools (a coupling channel is processed before its targets,
 * spectral_to_sample walks the types downwards); NULL / 0 for none.  The noise generator runs across the elements
 * of an access unit in bitstream order, so the caller orders the calls by HeaacCceFrame.behind_target:
 *   coupling element first:  tools_ex(cce, PRE | POST) -> tools_ex(target, PRE | POST, cce)
 *   target first:            tools_ex(target, PRE) -> tools_ex(cce, PRE | POST) -> tools_ex(target, POST, cce)
 * with d_rng chained from call to call (d_rng / d_pred are only touched by PRE).  All frames of one call share the
 * order. */
enum { HEAAC_TOOLS_PRE = 1, HEAAC_TOOLS_POST = 2, HEAAC_TOOLS_ALL = 3 };
int heaac_spectral_tools_batch_ex(HeaacDevice *dev, int channels, int stages, float *d_coeffs,
                                  const HeaacToolsFrame *d_tools,
                                  const int32_t *d_rng_in, int32_t *d_rng_out,
                                  const HeaacPredictorState *d_pred_in, HeaacPredictorState *d_pred_out,
                                  const HeaacCceFrame *d_cce, const float *d_cce_coeffs, int n_cce,
                                  size_t n, void *stream);

/* Host-side helper (no GPU): derive the frequency-band tables of one SBR
 * header -- sbr_make_f_master/f_derived/hf_calc_npatches/f_tablelim
 * (aacsbr.c:146-205, 296-593).  sample_rate is the SBR (output) rate.
 * Returns 0, or HEAAC_ERR_ARG where the reference logs an error and falls
 * back to "pure upsampling mode" (aacsbr.c:1029-1033). */
int heaac_sbr_make_header(HeaacSbrHeader *h, int sample_rate,
                          int bs_start_freq, int bs_stop_freq, int bs_xover_band,
                          int bs_freq_scale, int bs_alter_scale, int bs_noise_bands,
                          int bs_limiter_bands, int bs_limiter_gains,
                          int bs_interpol_freq, int bs_smoothing_mode,
                          int bs_amp_res_header);

/* Library identification; also the "is the HIP code object present" probe. */
const char *heaac_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* HEAAC_DSP_H */
