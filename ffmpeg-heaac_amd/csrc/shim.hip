// shim.hip -- per-call compatibility surfaces on top of the batched kernels:
//   include/heaac_fft.h   (FFTContext / ff_mdct_init / ff_imdct_half ...)
//   include/heaac_codec.h (AVCodec-shaped decoder, one stream per context)
// Host pointers in, host pointers out; every call is a batch of one on the GPU.
// There is no CPU arithmetic path: without a usable device init fails (-1) and
// the transform entry points abort loudly.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "heaac_fft.h"
#include "heaac_codec.h"
#include "tables.h"
#include "kernels.h"
#include "validate.h"
#include "heaac_parse.h"

// capi.hip
extern "C" const float *heaac_device_tables(HeaacDevice *dev, const uint16_t **rev);

#include <pthread.h>

static HeaacDevice *g_dev;
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;     // g_dev and the side-record list

static void die(const char *what)
{
    fprintf(stderr, "heaac: %s failed and there is no CPU fallback\n", what);
    abort();
}

#define HIPCHK(x, what) do { if ((x) != hipSuccess) die(what); } while (0)

static HeaacDevice *default_device(void)
{
    pthread_mutex_lock(&g_lock);
    if (!g_dev) {
        const int rc = heaac_device_create(&g_dev, 64);
        if (rc != HEAAC_OK) {
            fprintf(stderr, "heaac: cannot create device context: %s\n", heaac_strerror(rc));
            g_dev = NULL;
        }
    }
    HeaacDevice *d = g_dev;
    pthread_mutex_unlock(&g_lock);
    return d;
}

// ---------------------------------------------------------------------------
// FFTContext surface
// ---------------------------------------------------------------------------
// struct FFTContext has no field for a backend's own data (fft.h:32-53), so each initialised
// context has a side record, found through the table pointer the context owns (revtab: one heap block
// per context, so the link also survives a by-value copy of the struct): which transform instance it is
// and the device buffers its calls reuse.
struct FftSide {
    const void *key;              // s->revtab
    int which;                    // MDCT instance 0..3 (heaac_dsp.h), -1 = plain FFT
    float *d_in, *d_out;          // device staging, sized for the transform
    FftSide *next;
};
static FftSide *g_sides;

static FftSide *side_of(const FFTContext *s)
{
    pthread_mutex_lock(&g_lock);
    FftSide *p = g_sides;
    while (p && p->key != s->revtab) p = p->next;
    pthread_mutex_unlock(&g_lock);
    return p;
}

static FftSide *side_add(const FFTContext *s, size_t in_bytes, size_t out_bytes)
{
    FftSide *p = (FftSide *)calloc(1, sizeof(*p));
    if (!p) return NULL;
    p->key = s->revtab;
    p->which = -1;
    if (hipMalloc((void **)&p->d_in, in_bytes) != hipSuccess ||
        (out_bytes && hipMalloc((void **)&p->d_out, out_bytes) != hipSuccess)) {
        if (p->d_in) (void)hipFree(p->d_in);
        free(p);
        return NULL;
    }
    pthread_mutex_lock(&g_lock);
    p->next = g_sides;
    g_sides = p;
    pthread_mutex_unlock(&g_lock);
    return p;
}

static void side_drop(const FFTContext *s)
{
    if (!s->revtab) return;
    pthread_mutex_lock(&g_lock);
    FftSide **pp = &g_sides;
    while (*pp && (*pp)->key != s->revtab) pp = &(*pp)->next;
    FftSide *p = *pp;
    if (p) *pp = p->next;
    pthread_mutex_unlock(&g_lock);
    if (p) {
        if (p->d_in) (void)hipFree(p->d_in);
        if (p->d_out) (void)hipFree(p->d_out);
        free(p);
    }
}

static void *aligned16(size_t bytes)
{
    void *p = NULL;                     // av_malloc: 16-byte aligned (libavutil/mem.c:83)
    return posix_memalign(&p, 16, bytes ? bytes : 16) == 0 ? p : NULL;
}

static void hip_fft_permute(FFTContext *s, FFTComplex *z)
{
    // ff_fft_permute_c (fft.c:180-202): pure data movement, stays on the host
    const int np = 1 << s->nbits;
    for (int j = 0; j < np; j++) s->tmp_buf[s->revtab[j]] = z[j];
    memcpy(z, s->tmp_buf, np * sizeof(FFTComplex));
}

static void hip_fft_calc(FFTContext *s, FFTComplex *z)
{
    HeaacDevice *dev = default_device();
    FftSide *sd = side_of(s);
    if (!dev || !sd) die("ff_fft_calc");
    const size_t bytes = sizeof(FFTComplex) << s->nbits;
    HIPCHK(hipMemcpy(sd->d_in, z, bytes, hipMemcpyHostToDevice), "hipMemcpy");
    if (heaac_launch_fft_calc(heaac_device_tables(dev, NULL), s->nbits, sd->d_in, 1, 0) != HEAAC_OK) die("ff_fft_calc");
    HIPCHK(hipMemcpy(z, sd->d_in, bytes, hipMemcpyDeviceToHost), "hipMemcpy");
}

static void hip_imdct(FFTContext *s, FFTSample *output, const FFTSample *input, int full)
{
    HeaacDevice *dev = default_device();
    FftSide *sd = side_of(s);
    if (!dev || !sd || sd->which < 0) die("ff_imdct_half");
    const int n = 1 << s->mdct_bits, n2 = n >> 1, n4 = n >> 2;
    // the FFT staging block (n/4 complex = n2 floats) takes the input, d_out the n outputs
    HIPCHK(hipMemcpy(sd->d_in, input, n2 * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy");
    if (heaac_imdct_half_batch(dev, sd->which, full ? sd->d_out + n4 : sd->d_out, sd->d_in, 1, NULL) != HEAAC_OK)
        die("ff_imdct_half");
    if (full && heaac_launch_imdct_mirror(sd->d_out, n, 1, 0) != HEAAC_OK) die("ff_imdct_calc");
    HIPCHK(hipMemcpy(output, sd->d_out, (full ? n : n2) * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy");
}

static void hip_imdct_half(FFTContext *s, FFTSample *o, const FFTSample *i) { hip_imdct(s, o, i, 0); }
static void hip_imdct_calc(FFTContext *s, FFTSample *o, const FFTSample *i) { hip_imdct(s, o, i, 1); }
static void hip_mdct_calc(FFTContext *s, FFTSample *o, const FFTSample *i)
{
    (void)s; (void)o; (void)i;
    die("ff_mdct_calc (forward MDCT is not on the decode path)");
}

extern "C" int ff_fft_init(FFTContext *s, int nbits, int inverse)
{
    // fft.c:81-176 with split_radix = 1; supported: the inverse FFTs of the path
    if (!s || !inverse || (nbits != 5 && nbits != 6 && nbits != 9 && nbits != 4))
        return -1;
    if (!default_device())
        return -1;
    const int n = 1 << nbits;
    s->nbits = nbits;
    s->inverse = inverse;
    s->exptab = NULL;
    s->exptab1 = NULL;
    s->revtab = (uint16_t *)aligned16(n * sizeof(uint16_t));
    s->tmp_buf = (FFTComplex *)aligned16(n * sizeof(FFTComplex));
    if (!s->revtab || !s->tmp_buf) { ff_fft_end(s); return -1; }
    HeaacHostTables *t = (HeaacHostTables *)malloc(sizeof(*t));
    if (!t) { ff_fft_end(s); return -1; }
    heaac_build_tables(t);
    if (nbits == 9)      memcpy(s->revtab, t->rev + RV_512, n * sizeof(uint16_t));
    else if (nbits == 6) memcpy(s->revtab, t->rev + RV_64, n * sizeof(uint16_t));
    else if (nbits == 5) memcpy(s->revtab, t->rev + RV_32, n * sizeof(uint16_t));
    else { free(t); ff_fft_end(s); return -1; }
    free(t);
    s->fft_permute = hip_fft_permute;
    s->fft_calc    = hip_fft_calc;
    s->imdct_calc  = hip_imdct_calc;
    s->imdct_half  = hip_imdct_half;
    s->mdct_calc   = hip_mdct_calc;
    s->split_radix = 1;
    // device staging reused by every call on this context: n complex in place for the FFT;
    // an MDCT over it (n4 = this n) needs 2 n floats in and 4 n floats out
    if (!side_add(s, (size_t)n * sizeof(FFTComplex), (size_t)n * 4 * sizeof(float))) { ff_fft_end(s); return -1; }
    return 0;
}

extern "C" void ff_fft_end(FFTContext *s)
{
    if (!s) return;
    side_drop(s);
    free(s->revtab);  s->revtab = NULL;
    free(s->tmp_buf); s->tmp_buf = NULL;
    s->exptab = NULL; s->exptab1 = NULL;
}

extern "C" void ff_fft_permute(FFTContext *s, FFTComplex *z) { s->fft_permute(s, z); }
extern "C" void ff_fft_calc(FFTContext *s, FFTComplex *z) { s->fft_calc(s, z); }

extern "C" int ff_mdct_init(FFTContext *s, int nbits, int inverse, double scale)
{
    // mdct.c:61-105
    if (!s) return -1;
    memset(s, 0, sizeof(*s));
    int which = -1;
    if (inverse && nbits == 11 && scale == 1.0) which = 0;
    else if (inverse && nbits == 8 && scale == 1.0) which = 1;
    else if (inverse && nbits == 7 && scale == 1.0 / 64) which = 2;
    else if (inverse && nbits == 7 && scale == -2.0) which = 3;
    if (which < 0) return -1;
    const int n = 1 << nbits, n4 = n >> 2;
    s->mdct_bits = nbits;
    s->mdct_size = n;
    s->permutation = FF_MDCT_PERM_NONE;
    if (ff_fft_init(s, nbits - 2, inverse) < 0) return -1;
    s->tcos = (FFTSample *)aligned16(n / 2 * sizeof(FFTSample));
    if (!s->tcos) { ff_mdct_end(s); return -1; }
    s->tsin = s->tcos + n4;
    HeaacHostTables *t = (HeaacHostTables *)malloc(sizeof(*t));
    if (!t) { ff_mdct_end(s); return -1; }
    heaac_build_tables(t);
    const int off = which == 0 ? TB_ROT2048 : which == 1 ? TB_ROT256 : which == 2 ? TB_ROT128S : TB_ROT128A;
    memcpy(s->tcos, t->f + off, n / 2 * sizeof(float));
    free(t);
    FftSide *sd = side_of(s);
    if (!sd) { ff_mdct_end(s); return -1; }
    sd->which = which;            // (the two N = 128 instances differ only in their tables)
    return 0;
}

extern "C" void ff_mdct_end(FFTContext *s)
{
    if (!s) return;
    free(s->tcos); s->tcos = NULL; s->tsin = NULL;
    ff_fft_end(s);
}

extern "C" void ff_imdct_half(FFTContext *s, FFTSample *o, const FFTSample *i) { s->imdct_half(s, o, i); }
extern "C" void ff_imdct_calc(FFTContext *s, FFTSample *o, const FFTSample *i) { s->imdct_calc(s, o, i); }

// Window generators are init-time host functions in the reference too.
extern "C" void ff_kbd_window_init(float *window, float alpha, int n)
{
    // mdct.c:35-54
    double sum = 0.0, *local = (double *)malloc(sizeof(double) * n);
    const double a2 = (alpha * M_PI / n) * (alpha * M_PI / n);
    for (int i = 0; i < n; i++) {
        const double tmp = i * (n - i) * a2;
        double bessel = 1.0;
        for (int j = 50; j > 0; j--) bessel = bessel * tmp / (j * j) + 1;
        sum += bessel;
        local[i] = sum;
    }
    sum++;
    for (int i = 0; i < n; i++) window[i] = (float)sqrt(local[i] / sum);
    free(local);
}

extern "C" void ff_sine_window_init(float *window, int n)
{
    for (int i = 0; i < n; i++)
        window[i] = sinf((float)((i + 0.5) * (M_PI / (2.0 * n))));
}

static float sine_128[128], sine_1024[1024];
extern "C" float *const ff_sine_windows[13] = {
    NULL, NULL, NULL, NULL, NULL, NULL, NULL, sine_128, NULL, NULL, sine_1024, NULL, NULL
};
extern "C" void ff_init_ff_sine_windows(int index)
{
    if (index >= 0 && index < 13 && ff_sine_windows[index])
        ff_sine_window_init(ff_sine_windows[index], 1 << index);
}

extern "C" FFTContext *av_mdct_init(int nbits, int inverse, double scale)
{
    FFTContext *s = (FFTContext *)aligned16(sizeof(*s));
    if (s && ff_mdct_init(s, nbits, inverse, scale) < 0) { free(s); s = NULL; }
    return s;
}
extern "C" void av_imdct_calc(FFTContext *s, FFTSample *o, const FFTSample *i) { s->imdct_calc(s, o, i); }
extern "C" void av_imdct_half(FFTContext *s, FFTSample *o, const FFTSample *i) { s->imdct_half(s, o, i); }
extern "C" void av_mdct_calc(FFTContext *s, FFTSample *o, const FFTSample *i) { s->mdct_calc(s, o, i); }
extern "C" void av_mdct_end(FFTContext *s) { if (s) { ff_mdct_end(s); free(s); } }

// avfft.c:25-51
extern "C" FFTContext *av_fft_init(int nbits, int inverse)
{
    FFTContext *s = (FFTContext *)aligned16(sizeof(*s));
    if (!s) return NULL;
    memset(s, 0, sizeof(*s));
    if (ff_fft_init(s, nbits, inverse) < 0) { free(s); return NULL; }   // (the reference returns a dead context here)
    return s;
}
extern "C" void av_fft_permute(FFTContext *s, FFTComplex *z) { s->fft_permute(s, z); }
extern "C" void av_fft_calc(FFTContext *s, FFTComplex *z) { s->fft_calc(s, z); }
extern "C" void av_fft_end(FFTContext *s) { if (s) { ff_fft_end(s); free(s); } }

// ---------------------------------------------------------------------------
// AVCodec-shaped decoder
// ---------------------------------------------------------------------------
#define MAX_HDRS 64
#include "codec_layout.h"

typedef struct HeaacDecoderPriv {
    HeaacDevice *dev;
    int cfg, ncore, nout, out_len;
    int downsampled;              // SBR with the output at the core rate (bitstream mode; publish_cfg)
    size_t words;
    float *d_state;               // one record, updated in place
    float *d_coeffs;
    uint8_t *d_side;              // ics | sbr | ps
    HeaacSbrHeader *d_hdr;
    int16_t *d_pcm;
    HeaacSbrHeader hdr[MAX_HDRS];
    // cfg = HEAAC_CFG_FROM_STREAM: the packets are access units
    int bitstream, configured, have_m4ac;
    HeaacAacConfig m4ac;
    HeaacAacStream ast;
    HeaacSbrStream sst;
    HeaacSbrHeaderTable *tab;
    size_t hdr_uploaded;
    HeaacToolsFrame *d_tools;
    int32_t *d_rng;
    HeaacPredictorState *d_pred;
    float *h_coeffs;
    HeaacToolsFrame *h_tools;
    // streams with several output elements per access unit (channel configurations 3..7, program config elements)
    int have_layout;
    HeaacAacLayout layout;
    struct HeaacLayoutDec *lay;
} HeaacDecoderPriv;

static void set_cfg(HeaacDecoderPriv *p, int cfg)
{
    p->cfg = cfg;
    switch (cfg) {
    case HEAAC_CFG_LC_MONO:   p->ncore = 1; p->nout = 1; p->out_len = 1024; p->words = HEAAC_STATE_WORDS_LC_MONO; break;
    case HEAAC_CFG_LC_STEREO: p->ncore = 2; p->nout = 2; p->out_len = 1024; p->words = HEAAC_STATE_WORDS_LC_STEREO; break;
    case HEAAC_CFG_HEV1:      p->ncore = 2; p->nout = 2; p->out_len = 2048; p->words = HEAAC_STATE_WORDS_HEV1; break;
    case HEAAC_CFG_HEV1_MONO: p->ncore = 1; p->nout = 1; p->out_len = 2048; p->words = HEAAC_STATE_WORDS_HEV1_MONO; break;
    case HEAAC_CFG_HEV2:      p->ncore = 1; p->nout = 2; p->out_len = 2048; p->words = HEAAC_STATE_WORDS_HEV2; break;
    default:                  p->ncore = 0; break;
    }
}

static int cfg_is_he(int cfg) { return cfg == HEAAC_CFG_HEV1 || cfg == HEAAC_CFG_HEV1_MONO || cfg == HEAAC_CFG_HEV2; }

// The two decisions the reference takes from the configuration's two sample rates: ff_sbr_apply synthesises with the
// 32-band bank -- 1024 samples per frame -- when ext_sample_rate < sbr->sample_rate = 2 * sample_rate (aacsbr.c:1055,
// :1719), and aac_decode_frame hands out 1024 << (ext_sample_rate > sample_rate) samples (aacdec.c:2080-2081).  No
// extension rate (implicit SBR) means twice the core rate (aacsbr.c:1056-1057).  Returns 1 for "downsampled SBR"
// (both say 1024), 0 for the usual case (both say 2048), -1 where they disagree (an extension rate strictly between:
// the reference would hand out 2048 samples of which the synthesis wrote 1024).
int heaac_sbr_output_mode(const HeaacAacConfig *m)
{
    const int ext = m->ext_sample_rate ? m->ext_sample_rate : 2 * m->sample_rate;
    const int downsampled = ext < 2 * m->sample_rate, doubled = ext > m->sample_rate;
    return downsampled == doubled ? -1 : downsampled;
}

// what the stream says so far -> the caller-visible fields (aacdec.c:2080-2094: samples = 1024 << multiplier)
static void publish_cfg(HeaacCodecContext *avctx, HeaacDecoderPriv *p)
{
    const int he = cfg_is_he(p->cfg);
    p->downsampled = he && p->bitstream && heaac_sbr_output_mode(&p->m4ac) == 1;
    if (he) p->out_len = p->downsampled ? 1024 : 2048;
    avctx->channels = p->nout;
    // output_configure (aacdec.c:247): aac_channel_layout[channel_config - 1] -- by the channel CONFIGURATION, so a mono
    // stream decoded with Parametric Stereo has two channels and says AV_CH_LAYOUT_MONO
    avctx->channel_layout = p->ncore == 2 ? HEAAC_CH_LAYOUT_STEREO : HEAAC_CH_LAYOUT_MONO;
    avctx->frame_size = p->out_len;
    avctx->sample_rate = he && !p->downsampled ? 2 * p->m4ac.sample_rate : p->m4ac.sample_rate;
}

static int dec_init_bitstream(HeaacCodecContext *avctx, HeaacDecoderPriv *p)
{
    p->bitstream = 1;
    if (avctx->extradata && avctx->extradata_size > 0) {
        // decode_audio_specific_config (aacdec.c:462-493): 960-sample frames are refused at init; the channel
        // configuration, or the program config element standing in for it, gives the output layout
        if (heaac_asc_layout(&p->m4ac, &p->layout, avctx->extradata, avctx->extradata_size) < 0) return -1;
        if (p->m4ac.object_type != HEAAC_AOT_AAC_LC && p->m4ac.object_type != HEAAC_AOT_AAC_MAIN) return -1;
        // one SCE or one CPE: the single-element path (with Parametric Stereo and coupling elements); else the layout path
        p->have_layout = p->m4ac.chan_config != 1 && p->m4ac.chan_config != 2;
        // A program-config layout with explicitly signalled SBR leaves ps = -1 (no channel count to rule it out,
        // mpeg4audio.c:137-139), which decode_audio_specific_config turns into ps = 1 (:476-477): the reference then gives
        // every SCE of the layout a second, Parametric Stereo output channel (che_configure, :203-206; codec_layout.hip).
        if (p->have_layout && p->m4ac.sbr == 1 && p->m4ac.ps == -1) p->m4ac.ps = 1;
        if (p->m4ac.sbr == 1 && heaac_sbr_output_mode(&p->m4ac) < 0) return -1;
        p->have_m4ac = 1;
    }
    const size_t words = HEAAC_STATE_WORDS_HEV2 > HEAAC_STATE_WORDS_HEV1 ? HEAAC_STATE_WORDS_HEV2 : HEAAC_STATE_WORDS_HEV1;
    p->tab = heaac_sbr_table_create(MAX_HDRS);
    p->h_coeffs = (float *)calloc(2 * 1024, sizeof(float));
    p->h_tools = (HeaacToolsFrame *)calloc(1, sizeof(HeaacToolsFrame));
    if (!p->tab || !p->h_coeffs || !p->h_tools) return -1;
    heaac_sbr_stream_init(&p->sst, 1);
    if (heaac_device_create(&p->dev, 64) != HEAAC_OK) return -1;
    if (hipMalloc((void **)&p->d_state, words * 4) != hipSuccess ||
        hipMalloc((void **)&p->d_coeffs, 2 * 1024 * 4) != hipSuccess ||
        hipMalloc((void **)&p->d_side, 2048) != hipSuccess ||
        hipMalloc((void **)&p->d_hdr, sizeof(p->hdr)) != hipSuccess ||
        hipMalloc((void **)&p->d_pcm, 2 * 2048 * 2) != hipSuccess ||
        hipMalloc((void **)&p->d_tools, sizeof(HeaacToolsFrame)) != hipSuccess ||
        hipMalloc((void **)&p->d_rng, 4) != hipSuccess ||
        hipMalloc((void **)&p->d_pred, 2 * HEAAC_MAX_PREDICTORS * sizeof(HeaacPredictorState)) != hipSuccess)
        return -1;
    if (hipMemset(p->d_state, 0, words * 4) != hipSuccess) return -1;
    const int32_t seed = 0x1f2e3d4c;                                   // ac->random_state, aacdec.c:558
    if (hipMemcpy(p->d_rng, &seed, 4, hipMemcpyHostToDevice) != hipSuccess) return -1;
    HeaacPredictorState *ps = (HeaacPredictorState *)calloc(2 * HEAAC_MAX_PREDICTORS, sizeof(*ps));
    if (!ps) return -1;
    for (int i = 0; i < 2 * HEAAC_MAX_PREDICTORS; i++) ps[i].var0 = ps[i].var1 = 1.0f;   // reset_predict_state, :507-515
    const hipError_t e = hipMemcpy(p->d_pred, ps, 2 * HEAAC_MAX_PREDICTORS * sizeof(*ps), hipMemcpyHostToDevice);
    free(ps);
    if (e != hipSuccess) return -1;
    if (p->have_m4ac && p->have_layout) {
        if (!(p->lay = heaac_layout_dec_create(p->dev, &p->m4ac, &p->layout))) return -1;
        // tentative (output_configure with OC_GLOBAL_HDR); the first access unit settles implicit SBR
        const int doubled = p->m4ac.sbr == 1 && heaac_sbr_output_mode(&p->m4ac) == 0;
        avctx->channels = heaac_layout_dec_channels(p->lay);
        avctx->channel_layout = p->layout.channel_layout;
        avctx->frame_size = doubled ? 2048 : 1024;
        avctx->sample_rate = doubled ? 2 * p->m4ac.sample_rate : p->m4ac.sample_rate;
    } else if (p->have_m4ac) {
        // tentative, as decode_audio_specific_config leaves it; the first access unit settles implicit SBR
        const int he = p->m4ac.sbr == 1;
        set_cfg(p, he ? (p->m4ac.chan_config == 2 ? HEAAC_CFG_HEV1 : (p->m4ac.ps != 0 ? HEAAC_CFG_HEV2 : HEAAC_CFG_HEV1_MONO))
                      : (p->m4ac.chan_config == 2 ? HEAAC_CFG_LC_STEREO : HEAAC_CFG_LC_MONO));
        publish_cfg(avctx, p);
    }
    return 0;
}

static int dec_frame_bitstream(HeaacCodecContext *avctx, HeaacDecoderPriv *p, void *data, int *data_size,
                               HeaacPacket *avpkt)
{
    const uint8_t *buf = avpkt->data;
    const int size = avpkt->size;
    if (!buf || size < 2) return -1;
    if (!p->have_m4ac) {
        // parse_adts_frame_header (aacdec.c:1935-1971): the stream configures itself
        HeaacAdtsHeader ah;
        const int hs = heaac_adts_parse_header(&ah, buf, size);
        if (hs < 0) return -1;
        memset(&p->m4ac, 0, sizeof(p->m4ac));
        p->m4ac.object_type = ah.object_type;
        p->m4ac.sampling_index = ah.sampling_index;
        p->m4ac.sample_rate = ah.sample_rate;
        p->m4ac.chan_config = ah.chan_config;
        p->m4ac.sbr = -1;
        p->m4ac.ps = -1;
        // The header's two profile bits give object types 1 .. 4, and unlike decode_audio_specific_config (:481-491)
        // this way in asks nothing of them: an SSR or LTP profile stream decodes as AAC-LC does until an element uses
        // what the reference lacks (gain control :1373, the predictor bit :694), which fails that frame.
        if (ah.chan_config != 1 && ah.chan_config != 2) {
            // set_default_channel_config (:1946), or -- channel configuration 0 -- the program config element ahead of
            // the raw data block's channel elements (:2036-2046, OC_TRIAL_PCE)
            if (ah.chan_config) {
                if (heaac_aac_layout_default(&p->layout, ah.chan_config) < 0) return -1;
            } else {
                if (heaac_aac_layout_from_au(&p->layout, buf, size) < 0) return -1;
            }
            if (!(p->lay = heaac_layout_dec_create(p->dev, &p->m4ac, &p->layout))) return -1;
            p->have_layout = 1;
        }
        p->have_m4ac = 1;
    }
    if (p->have_layout) {
        HeaacLayoutOut lo;
        const int used = heaac_layout_dec_frame(p->lay, buf, size, data, data_size, &lo);
        if (used < 0) return -1;
        if (!p->configured) {
            avctx->channels = lo.channels;
            avctx->channel_layout = lo.channel_layout;
            avctx->frame_size = lo.frame_size;
            avctx->sample_rate = lo.sample_rate;
            p->configured = 1;
        }
        return used;
    }
    HeaacIcs ics[2];
    HeaacSbrFrame sbr;
    HeaacPsFrame ps;
    HeaacAacFrameInfo fi;
    memset(ics, 0, sizeof(ics));
    const int r = heaac_heaac_parse_frame(&p->m4ac, &p->ast, &p->sst, p->tab, buf, size, p->h_coeffs, ics, p->h_tools,
                                          &sbr, &ps, &fi);
    // A coupling channel element lands here too: in a channel configuration 1 / 2 stream get_che has no place for it
    // ("channel element 2.%d is not allocated", aacdec.c:132-177, :2006-2010) -- only a program config element
    // allocates coupling elements, and those streams take the layout path above.
    if (r < 0 && fi.channels == 0 && (fi.refused & HEAAC_REFUSED_RUN_TOOLS) && p->configured) {
        // No samples, but the reference's element decoders had drawn noise / stepped predictors before they refused
        // the unit (heaac_parse.h, HEAAC_REFUSED_*): the records the parser left do exactly that much to the stream's
        // generator and predictors; the coefficients they leave are dropped, the decoder's state stays.
        const int ch = p->ncore;
        const int main_profile = p->m4ac.object_type == HEAAC_AOT_AAC_MAIN;
        if (hipMemcpy(p->d_coeffs, p->h_coeffs, (size_t)ch * 4096, hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(p->d_tools, p->h_tools, sizeof(HeaacToolsFrame), hipMemcpyHostToDevice) == hipSuccess &&
            heaac_spectral_tools_batch(p->dev, ch, p->d_coeffs, p->d_tools, p->d_rng, p->d_rng,
                                       main_profile ? p->d_pred : NULL, main_profile ? p->d_pred : NULL, 1, NULL) == HEAAC_OK)
            (void)hipDeviceSynchronize();
        return -1;
    }
    if (r == HEAAC_PARSE_ERR_UNSUPPORTED) return -1;
    if (r == HEAAC_PARSE_ERR_ARG) return -1;
    if (r < 0 && fi.channels == 0) return -1;          // the core element failed (aacdec.c:2046-2049)
    if (fi.channels != p->m4ac.chan_config) return -1;
    if (!p->configured) {
        // implicit SBR counts only when the first access unit carries it (aacdec.c:1666-1675)
        if (p->m4ac.sbr == -1) p->m4ac.sbr = fi.sbr_payload_bit >= 0 ? 1 : 0;
        const int he = p->m4ac.sbr == 1;
        set_cfg(p, he ? (fi.channels == 2 ? HEAAC_CFG_HEV1 : (p->m4ac.ps != 0 ? HEAAC_CFG_HEV2 : HEAAC_CFG_HEV1_MONO))
                      : (fi.channels == 2 ? HEAAC_CFG_LC_STEREO : HEAAC_CFG_LC_MONO));
        publish_cfg(avctx, p);
        p->configured = 1;
    }
    const int he = cfg_is_he(p->cfg);
    const int ch = fi.channels;
    uint8_t *d_ics = p->d_side, *d_sbr = p->d_side + 16, *d_ps = p->d_side + 16 + 688;
    if (hipMemcpy(p->d_coeffs, p->h_coeffs, (size_t)ch * 4096, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_ics, ics, sizeof(ics), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(p->d_tools, p->h_tools, sizeof(HeaacToolsFrame), hipMemcpyHostToDevice) != hipSuccess)
        return -1;
    const int main_profile = p->m4ac.object_type == HEAAC_AOT_AAC_MAIN;
    int rc = heaac_spectral_tools_batch(p->dev, ch, p->d_coeffs, p->d_tools, p->d_rng, p->d_rng,
                                        main_profile ? p->d_pred : NULL, main_profile ? p->d_pred : NULL, 1, NULL);
    if (rc != HEAAC_OK) return -1;
    if (!he) {
        rc = heaac_lc_decode_batch(p->dev, ch, p->d_coeffs, (const HeaacIcs *)d_ics, p->d_state, p->d_state,
                                   p->d_pcm, HEAAC_PCM_S16_INTERLEAVED, 1, NULL);
    } else {
        const size_t have = heaac_sbr_table_count(p->tab);
        if (have > p->hdr_uploaded) {
            memcpy(p->hdr + p->hdr_uploaded, heaac_sbr_table_data(p->tab) + p->hdr_uploaded,
                   (have - p->hdr_uploaded) * sizeof(HeaacSbrHeader));
            if (hipMemcpy(p->d_hdr + p->hdr_uploaded, p->hdr + p->hdr_uploaded,
                          (have - p->hdr_uploaded) * sizeof(HeaacSbrHeader), hipMemcpyHostToDevice) != hipSuccess)
                return -1;
            p->hdr_uploaded = have;
        }
        if (heaac_validate_frame(p->cfg, &sbr, p->hdr, MAX_HDRS, p->cfg == HEAAC_CFG_HEV2 ? &ps : NULL)) return -1;
        if (hipMemcpy(d_sbr, &sbr, sizeof(sbr), hipMemcpyHostToDevice) != hipSuccess) return -1;
        if (p->cfg == HEAAC_CFG_HEV2 && hipMemcpy(d_ps, &ps, sizeof(ps), hipMemcpyHostToDevice) != hipSuccess) return -1;
        rc = heaac_he_decode_batch_ex(p->dev, p->cfg, p->downsampled ? HEAAC_HE_DOWNSAMPLED : 0, p->d_coeffs,
                                      (const HeaacIcs *)d_ics, (const HeaacSbrFrame *)d_sbr,
                                      p->d_hdr, MAX_HDRS, p->cfg == HEAAC_CFG_HEV2 ? (const HeaacPsFrame *)d_ps : NULL,
                                      p->d_state, p->d_state, p->d_pcm, HEAAC_PCM_S16_INTERLEAVED, 1, NULL);
    }
    if (rc != HEAAC_OK) return -1;
    const int bytes = p->out_len * p->nout * 2;
    if (*data_size < bytes) return -1;                       // "Output buffer too small" (aacdec.c:2087-2092)
    if (hipMemcpy(data, p->d_pcm, bytes, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    *data_size = bytes;
    // aacdec.c:2102-2107: bytes consumed, or the whole packet when only zero padding follows
    const int consumed = (fi.bits_consumed + 7) >> 3;
    int off = consumed;
    while (off < size && !buf[off]) off++;
    return size > off ? consumed : size;
}

static int dec_init(HeaacCodecContext *avctx)
{
    HeaacDecoderPriv *p = (HeaacDecoderPriv *)avctx->priv_data;
    avctx->sample_fmt = HEAAC_SAMPLE_FMT_S16;                          // aacdec.c:568
    if (!HEAAC_SUBID_IS_RECORDS(avctx->sub_id)) return dec_init_bitstream(avctx, p);
    set_cfg(p, avctx->sub_id & 0xff);
    if (!p->ncore) return -1;
    if (heaac_device_create(&p->dev, 64) != HEAAC_OK) return -1;
    if (hipMalloc((void **)&p->d_state, p->words * 4) != hipSuccess ||
        hipMalloc((void **)&p->d_coeffs, 2 * 1024 * 4) != hipSuccess ||
        hipMalloc((void **)&p->d_side, 2048) != hipSuccess ||
        hipMalloc((void **)&p->d_hdr, sizeof(p->hdr)) != hipSuccess ||
        hipMalloc((void **)&p->d_pcm, 2 * 2048 * 2) != hipSuccess)
        return -1;
    if (hipMemset(p->d_state, 0, p->words * 4) != hipSuccess) return -1;
    memset(p->hdr, 0, sizeof(p->hdr));
    for (int i = 0; i < MAX_HDRS; i++) p->hdr[i].kx = 32;      // kx' = 32, m = 0 (aacsbr.c:130)
    if (hipMemcpy(p->d_hdr, p->hdr, sizeof(p->hdr), hipMemcpyHostToDevice) != hipSuccess) return -1;
    avctx->channels = p->nout;
    avctx->channel_layout = p->ncore == 2 ? HEAAC_CH_LAYOUT_STEREO : HEAAC_CH_LAYOUT_MONO;   // as publish_cfg
    avctx->frame_size = p->out_len;
    if (!avctx->sample_rate) avctx->sample_rate = 48000;
    return 0;
}

static int dec_close(HeaacCodecContext *avctx)
{
    HeaacDecoderPriv *p = (HeaacDecoderPriv *)avctx->priv_data;
    if (!p) return 0;
    if (p->d_state) (void)hipFree(p->d_state);
    if (p->d_coeffs) (void)hipFree(p->d_coeffs);
    if (p->d_side) (void)hipFree(p->d_side);
    if (p->d_hdr) (void)hipFree(p->d_hdr);
    if (p->d_pcm) (void)hipFree(p->d_pcm);
    if (p->d_tools) (void)hipFree(p->d_tools);
    if (p->d_rng) (void)hipFree(p->d_rng);
    if (p->d_pred) (void)hipFree(p->d_pred);
    heaac_sbr_table_destroy(p->tab);
    heaac_layout_dec_destroy(p->lay);
    free(p->h_coeffs);
    free(p->h_tools);
    heaac_device_destroy(p->dev);
    memset(p, 0, sizeof(*p));
    return 0;
}

static int dec_frame(HeaacCodecContext *avctx, void *data, int *data_size, HeaacPacket *avpkt)
{
    HeaacDecoderPriv *p = (HeaacDecoderPriv *)avctx->priv_data;
    if (p->bitstream) return dec_frame_bitstream(avctx, p, data, data_size, avpkt);
    const uint8_t *buf = avpkt->data;
    const int he = p->cfg == HEAAC_CFG_HEV1 || p->cfg == HEAAC_CFG_HEV1_MONO || p->cfg == HEAAC_CFG_HEV2;
    size_t need = sizeof(HeaacFramePacket) + (size_t)p->ncore * 4096 + (he ? sizeof(HeaacSbrFrame) : 0) +
                  (p->cfg == HEAAC_CFG_HEV2 ? sizeof(HeaacPsFrame) : 0);
    if (!buf || (size_t)avpkt->size < need) return -1;
    HeaacFramePacket hp;
    memcpy(&hp, buf, sizeof(hp));
    if (hp.magic != HEAAC_PACKET_MAGIC || hp.cfg != p->cfg) return -1;
    const uint8_t *q = buf + sizeof(hp);
    const float *coeffs = (const float *)q;           q += (size_t)p->ncore * 4096;
    HeaacSbrFrame sbr; HeaacPsFrame ps;
    if (he) { memcpy(&sbr, q, sizeof(sbr)); q += sizeof(sbr); }
    if (p->cfg == HEAAC_CFG_HEV2) { memcpy(&ps, q, sizeof(ps)); q += sizeof(ps); }
    if (hp.flags & HEAAC_PKT_NEW_SBR_HEADER) {
        need += sizeof(HeaacSbrHeader);
        if ((size_t)avpkt->size < need || !he || sbr.hdr >= MAX_HDRS) return -1;
        HeaacSbrHeader nh;
        memcpy(&nh, q, sizeof(nh));
        if (heaac_check_sbr_header(&nh)) return -1;          // validate before it reaches the device table
        p->hdr[sbr.hdr] = nh;
        if (hipMemcpy(p->d_hdr + sbr.hdr, &p->hdr[sbr.hdr], sizeof(HeaacSbrHeader),
                      hipMemcpyHostToDevice) != hipSuccess) return -1;
    }
    if (he && sbr.hdr >= MAX_HDRS) return -1;
    // one frame: the host check is free (the reference rejects these while parsing, aacsbr.c:609-745)
    if (he && heaac_validate_frame(p->cfg, &sbr, p->hdr, MAX_HDRS, p->cfg == HEAAC_CFG_HEV2 ? &ps : NULL)) return -1;

    uint8_t *d_ics = p->d_side, *d_sbr = p->d_side + 16, *d_ps = p->d_side + 16 + 688;
    if (hipMemcpy(p->d_coeffs, coeffs, (size_t)p->ncore * 4096, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_ics, hp.ics, sizeof(hp.ics), hipMemcpyHostToDevice) != hipSuccess)
        return -1;
    int rc;
    if (!he) {
        rc = heaac_lc_decode_batch(p->dev, p->ncore, p->d_coeffs, (const HeaacIcs *)d_ics, p->d_state,
                                   p->d_state, p->d_pcm, HEAAC_PCM_S16_INTERLEAVED, 1, NULL);
    } else {
        if (hipMemcpy(d_sbr, &sbr, sizeof(sbr), hipMemcpyHostToDevice) != hipSuccess) return -1;
        if (p->cfg == HEAAC_CFG_HEV2 &&
            hipMemcpy(d_ps, &ps, sizeof(ps), hipMemcpyHostToDevice) != hipSuccess) return -1;
        rc = heaac_he_decode_batch(p->dev, p->cfg, p->d_coeffs, (const HeaacIcs *)d_ics,
                                   (const HeaacSbrFrame *)d_sbr, p->d_hdr, MAX_HDRS,
                                   p->cfg == HEAAC_CFG_HEV2 ? (const HeaacPsFrame *)d_ps : NULL,
                                   p->d_state, p->d_state, p->d_pcm, HEAAC_PCM_S16_INTERLEAVED, 1, NULL);
    }
    if (rc != HEAAC_OK) return -1;
    // *data_size = samples * channels * sizeof(int16_t) (aacdec.c:2087-2094)
    const int bytes = p->out_len * p->nout * 2;
    if (*data_size < bytes) return -1;                       // "Output buffer too small" (aacdec.c:2087-2092)
    if (hipMemcpy(data, p->d_pcm, bytes, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    *data_size = bytes;
    return (int)need;            // bytes consumed (aacdec.c:2102-2107)
}

// aacdec.c:2128-2142 (channel layouts: aac_channel_layout[], aacdectab.h:84-93)
static const int dec_sample_fmts[] = { HEAAC_SAMPLE_FMT_S16, HEAAC_SAMPLE_FMT_NONE };
static const int64_t dec_channel_layouts[] = { HEAAC_CH_LAYOUT_MONO, HEAAC_CH_LAYOUT_STEREO, 0x7, 0x107, 0x37, 0x3f, 0xff, 0 };   // aac_channel_layout[], aacdectab.h:84-93
extern "C" HeaacCodec heaac_aac_decoder = {
    "aac", 1, HEAAC_CODEC_ID_AAC, (int)sizeof(HeaacDecoderPriv), dec_init, NULL, dec_close, dec_frame,
    0, NULL, NULL, NULL, NULL, "Advanced Audio Coding (HE-AAC DSP on gfx950)", NULL,
    dec_sample_fmts, dec_channel_layouts,
};

extern "C" void heaac_codec_get_context_defaults(HeaacCodecContext *avctx)
{
    // avcodec_get_context_defaults2(s, AVMEDIA_TYPE_UNKNOWN), options.c: everything the decoder looks at is 0
    // except the media type
    memset(avctx, 0, sizeof(*avctx));
    avctx->codec_type = HEAAC_MEDIA_TYPE_UNKNOWN;
}

extern "C" int heaac_codec_open(HeaacCodecContext *avctx, HeaacCodec *codec)
{
    // utils.c:462-531
    if (!avctx || !codec || avctx->codec) return -1;
    avctx->priv_data = calloc(1, codec->priv_data_size);
    if (!avctx->priv_data) return -12;       /* AVERROR(ENOMEM) */
    avctx->codec = codec;
    // utils.c:506-514: an unset type / id takes the codec's, a different one is refused
    if ((avctx->codec_type == HEAAC_MEDIA_TYPE_UNKNOWN || avctx->codec_type == codec->type) && avctx->codec_id == 0) {
        avctx->codec_type = codec->type;
        avctx->codec_id = codec->id;
    }
    if (avctx->codec_id != codec->id || avctx->codec_type != codec->type) {
        free(avctx->priv_data);
        avctx->priv_data = NULL;
        avctx->codec = NULL;
        return -1;
    }
    avctx->frame_number = 0;
    const int ret = codec->init(avctx);
    if (ret < 0) {
        codec->close(avctx);
        free(avctx->priv_data);
        avctx->priv_data = NULL;
        avctx->codec = NULL;
    }
    return ret;
}

extern "C" int heaac_codec_decode(HeaacCodecContext *avctx, int16_t *samples, int *frame_size_ptr,
                                  HeaacPacket *avpkt)
{
    // utils.c:638-663
    if (!avctx || !avctx->codec || !samples || !frame_size_ptr || !avpkt) return -1;
    if (avpkt->size) {
        if (*frame_size_ptr < HEAAC_MAX_AUDIO_FRAME_SIZE) return -1;
        const int ret = avctx->codec->decode(avctx, samples, frame_size_ptr, avpkt);
        avctx->frame_number++;                                         // utils.c:655
        return ret;
    }
    *frame_size_ptr = 0;
    return 0;
}

extern "C" int heaac_codec_close(HeaacCodecContext *avctx)
{
    if (!avctx) return -1;
    if (avctx->codec && avctx->codec->close) avctx->codec->close(avctx);
    free(avctx->priv_data);
    avctx->priv_data = NULL;
    avctx->codec = NULL;
    return 0;
}
