/* abi_check.c -- the C ABI exercised from plain C (gcc), the way a host decoder would link it:
 *   abi_check cpu   layouts, host-side entry points, loud failure without a device
 *   abi_check gpu   transform plugin surface against O(N^2) double-precision references (the
 *                   reference's own fft-test.c criterion: every |err| < 1e-3), av_fft_*, two N = 128
 *                   MDCT contexts alive at once, the AVCodec-shaped decoder on AAC-LC packets
 * Exit code 0 = all checks passed.  Test infrastructure: links the product only.
 */
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "heaac_dsp.h"
#include "heaac_fft.h"
#include "heaac_codec.h"

static int fails;
#define CHECK(c) do { if (!(c)) { printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); fails++; } } while (0)

/* what struct AVCodec looks like on LP64 (avcodec.h:2675-2711) */
struct ref_AVCodec {
    const char *name; int type; int id; int priv_data_size;
    int (*init)(void *); int (*encode)(void *, uint8_t *, int, void *); int (*close)(void *);
    int (*decode)(void *, void *, int *, void *);
    int capabilities; struct ref_AVCodec *next; void (*flush)(void *);
    const void *supported_framerates; const int *pix_fmts; const char *long_name;
    const int *supported_samplerates; const int *sample_fmts; const int64_t *channel_layouts;
};

/* what struct AVPacket looks like (avcodec.h:960-1002), field for field */
struct ref_AVPacket {
    int64_t pts, dts; uint8_t *data; int size, stream_index, flags, duration;
    void (*destruct)(struct ref_AVPacket *); void *priv; int64_t pos, convergence_duration;
};
/* the head of struct AVCodecContext of libavcodec 52.78 (avcodec.h:1032-1293: every member up to codec_id, in
 * order, LIBAVCODEC_VERSION_MAJOR < 53 members included); the members behind it are not declared: channel_layout
 * (:2502) and the size are pinned as the numbers a build of that header gives on LP64 (976, 1088). */
struct ref_AVRational { int num, den; };
struct ref_AVCodecContext_head {
    const void *av_class; int bit_rate, bit_rate_tolerance, flags, sub_id, me_method;
    uint8_t *extradata; int extradata_size; struct ref_AVRational time_base; int width, height, gop_size;
    int pix_fmt, rate_emu; void (*draw_horiz_band)(void *, const void *, int *, int, int, int);
    int sample_rate, channels, sample_fmt, frame_size, frame_number, real_pict_num, delay;
    float qcompress, qblur; int qmin, qmax, max_qdiff, max_b_frames; float b_quant_factor;
    int rc_strategy, b_frame_strategy, hurry_up; struct ref_AVCodec *codec; void *priv_data;
    int rtp_payload_size; void (*rtp_callback)(void *, void *, int, int);
    int mv_bits, header_bits, i_tex_bits, p_tex_bits, i_count, p_count, skip_count, misc_bits, frame_bits;
    void *opaque; char codec_name[32]; int codec_type, codec_id;
};
#define REF_AVCTX_SIZE 1088
#define REF_AVCTX_CHANNEL_LAYOUT 976
union ref_AVCodecContext { struct ref_AVCodecContext_head h; uint8_t bytes[REF_AVCTX_SIZE]; int64_t align_; };

static unsigned lfg;                      /* any deterministic generator */
static float frand(void) { lfg = lfg * 1664525u + 1013904223u; return (int)(lfg >> 8) / 8388608.0f - 1.0f; }

static void imdct_ref(double *out, const float *in, int nbits, double scale)
{
    /* fft-test.c:98-113 with the transform's scale */
    const int n = 1 << nbits;
    for (int i = 0; i < n; i++) {
        double sum = 0;
        for (int k = 0; k < n / 2; k++) {
            const int a = (2 * i + 1 + (n / 2)) * (2 * k + 1);
            sum += cos((2 * M_PI * a) / (4.0 * n)) * in[k];
        }
        out[i] = -sum * scale;
    }
}

static int cpu_checks(void)
{
    CHECK(sizeof(HeaacCodec) == sizeof(struct ref_AVCodec));
    CHECK(offsetof(HeaacCodec, decode) == offsetof(struct ref_AVCodec, decode));
    CHECK(offsetof(HeaacCodec, capabilities) == offsetof(struct ref_AVCodec, capabilities));
    CHECK(offsetof(HeaacCodec, long_name) == offsetof(struct ref_AVCodec, long_name));
    CHECK(offsetof(HeaacCodec, sample_fmts) == offsetof(struct ref_AVCodec, sample_fmts));
    CHECK(offsetof(HeaacCodec, channel_layouts) == offsetof(struct ref_AVCodec, channel_layouts));
    /* the callbacks receive libavcodec's own AVCodecContext / AVPacket */
#define SAME(T, R, f) CHECK(offsetof(T, f) == offsetof(R, f) && sizeof(((T *)0)->f) == sizeof(((R *)0)->f))
    CHECK(sizeof(HeaacPacket) == sizeof(struct ref_AVPacket));
    SAME(HeaacPacket, struct ref_AVPacket, pts); SAME(HeaacPacket, struct ref_AVPacket, dts);
    SAME(HeaacPacket, struct ref_AVPacket, data); SAME(HeaacPacket, struct ref_AVPacket, size);
    SAME(HeaacPacket, struct ref_AVPacket, flags); SAME(HeaacPacket, struct ref_AVPacket, destruct);
    SAME(HeaacPacket, struct ref_AVPacket, pos); SAME(HeaacPacket, struct ref_AVPacket, convergence_duration);
    SAME(HeaacCodecContext, struct ref_AVCodecContext_head, av_class);
    SAME(HeaacCodecContext, struct ref_AVCodecContext_head, sub_id);
    SAME(HeaacCodecContext, struct ref_AVCodecContext_head, extradata);
    SAME(HeaacCodecContext, struct ref_AVCodecContext_head, extradata_size);
    SAME(HeaacCodecContext, struct ref_AVCodecContext_head, draw_horiz_band);
    SAME(HeaacCodecContext, struct ref_AVCodecContext_head, sample_rate);
    SAME(HeaacCodecContext, struct ref_AVCodecContext_head, channels);
    SAME(HeaacCodecContext, struct ref_AVCodecContext_head, sample_fmt);
    SAME(HeaacCodecContext, struct ref_AVCodecContext_head, frame_size);
    SAME(HeaacCodecContext, struct ref_AVCodecContext_head, frame_number);
    SAME(HeaacCodecContext, struct ref_AVCodecContext_head, codec);
    SAME(HeaacCodecContext, struct ref_AVCodecContext_head, priv_data);
    SAME(HeaacCodecContext, struct ref_AVCodecContext_head, codec_type);
    SAME(HeaacCodecContext, struct ref_AVCodecContext_head, codec_id);
    CHECK(offsetof(HeaacCodecContext, channel_layout) == REF_AVCTX_CHANNEL_LAYOUT);
    CHECK(sizeof(HeaacCodecContext) == REF_AVCTX_SIZE && sizeof(union ref_AVCodecContext) == REF_AVCTX_SIZE);
    CHECK(!strcmp(heaac_aac_decoder.name, "aac") && heaac_aac_decoder.type == 1);
    CHECK(heaac_aac_decoder.sample_fmts[0] == HEAAC_SAMPLE_FMT_S16 && heaac_aac_decoder.sample_fmts[1] == -1);
    {
        /* .channel_layouts = aac_channel_layout (aacdec.c:2140; aacdectab.h:84-93 with avcodec.h:386-427): mono, stereo,
         * 3.0, 4.0, 5.0 (back), 5.1 (back), 7.1 (wide), 0 */
        static const int64_t want[8] = { 0x4, 0x3, 0x7, 0x107, 0x37, 0x3f, 0xff, 0 };
        for (int i = 0; i < 8; i++) CHECK(heaac_aac_decoder.channel_layouts[i] == want[i]);
    }
    CHECK(sizeof(HeaacSbrHeader) == 532 && sizeof(HeaacSbrFrame) == 680 && sizeof(HeaacPsFrame) == 532);

    HeaacSbrHeader h;
    CHECK(heaac_sbr_make_header(&h, 48000, 5, 9, 0, 2, 1, 2, 2, 2, 1, 1, 1) == HEAAC_OK);
    CHECK(h.k0 == 13 && h.kx == 13 && h.m == 32 && h.n[1] == 16 && h.n[0] == 8 && h.n_q == 4 && h.num_patches == 3);
    HeaacSbrFrame f; memset(&f, 0, sizeof(f));
    f.kx_old = 32;
    CHECK(heaac_validate_frame(HEAAC_CFG_HEV1_MONO, &f, &h, 1, NULL) == HEAAC_BAD_NONE);      /* start = 0 */
    f.start = 1;
    CHECK(heaac_validate_frame(HEAAC_CFG_HEV1_MONO, &f, &h, 1, NULL) == HEAAC_BAD_SBR_NUM_ENV);
    f.hdr = 1;
    CHECK(heaac_validate_frame(HEAAC_CFG_HEV1_MONO, &f, &h, 1, NULL) == HEAAC_BAD_HDR_INDEX);
    CHECK(heaac_strerror(HEAAC_ERR_NODEVICE) != NULL);
    return fails;
}

static int no_device_checks(void)
{
    /* no GPU: nothing computes on the CPU instead */
    HeaacDevice *d = (HeaacDevice *)1;
    FFTContext c;
    CHECK(heaac_device_create(&d, 16) == HEAAC_ERR_NODEVICE && d == NULL);
    CHECK(ff_mdct_init(&c, 11, 1, 1.0) == -1);
    CHECK(av_fft_init(9, 1) == NULL);
    HeaacCodecContext ctx; heaac_codec_get_context_defaults(&ctx); ctx.sub_id = HEAAC_SUBID_RECORDS(HEAAC_CFG_LC_STEREO);
    CHECK(heaac_codec_open(&ctx, &heaac_aac_decoder) < 0 && ctx.priv_data == NULL);
    /* a context of another codec is refused before anything is allocated (utils.c:510-513) */
    heaac_codec_get_context_defaults(&ctx); ctx.codec_id = 0x15001;            /* CODEC_ID_MP3 */
    CHECK(heaac_codec_open(&ctx, &heaac_aac_decoder) < 0 && ctx.priv_data == NULL && ctx.codec == NULL);
    return fails;
}

static void mdct_case(int nbits, double scale)
{
    const int n = 1 << nbits;
    FFTContext c;
    float *in = malloc(n / 2 * sizeof(float)), *out = malloc(n * sizeof(float));
    double *ref = malloc(n * sizeof(double));
    CHECK(ff_mdct_init(&c, nbits, 1, scale) == 0);
    for (int i = 0; i < n / 2; i++) in[i] = frand();
    imdct_ref(ref, in, nbits, scale);
    ff_imdct_calc(&c, out, in);
    double worst = 0;
    for (int i = 0; i < n; i++) worst = fmax(worst, fabs(out[i] - ref[i]) / fabs(scale));
    CHECK(worst < 1e-3);
    ff_imdct_half(&c, out, in);           /* the middle half, mdct.c:124-159 */
    worst = 0;
    for (int i = 0; i < n / 2; i++) worst = fmax(worst, fabs(out[i] - ref[n / 4 + i]) / fabs(scale));
    CHECK(worst < 1e-3);
    printf("imdct N=%d scale %g: max err %.2e\n", n, scale, worst);
    ff_mdct_end(&c);
    free(in); free(out); free(ref);
}

static int gpu_checks(void)
{
    lfg = 1;
    mdct_case(11, 1.0); mdct_case(8, 1.0); mdct_case(7, 1.0 / 64); mdct_case(7, -2.0);

    /* two N = 128 contexts alive together keep their own tables (side records, not a shared tag) */
    FFTContext a, b;
    float in[64], oa[64], ob[64];
    CHECK(ff_mdct_init(&a, 7, 1, 1.0 / 64) == 0 && ff_mdct_init(&b, 7, 1, -2.0) == 0);
    for (int i = 0; i < 64; i++) in[i] = frand();
    ff_imdct_half(&b, ob, in); ff_imdct_half(&a, oa, in); ff_imdct_half(&b, ob, in);
    double ra[128], rb[128], wa = 0, wb = 0;
    imdct_ref(ra, in, 7, 1.0 / 64); imdct_ref(rb, in, 7, -2.0);
    for (int i = 0; i < 64; i++) { wa = fmax(wa, fabs(oa[i] - ra[32 + i]) * 64); wb = fmax(wb, fabs(ob[i] - rb[32 + i]) / 2); }
    CHECK(wa < 1e-3 && wb < 1e-3);
    ff_mdct_end(&a); ff_mdct_end(&b);

    /* av_fft_*: inverse complex FFT, 512 points, against the O(N^2) sum (fft-test.c:60-85) */
    FFTContext *s = av_fft_init(9, 1);
    CHECK(s != NULL);
    if (s) {
        static FFTComplex z[512], z0[512];
        for (int i = 0; i < 512; i++) { z[i].re = z0[i].re = frand(); z[i].im = z0[i].im = frand(); }
        av_fft_permute(s, z); av_fft_calc(s, z);
        double worst = 0;
        for (int i = 0; i < 512; i += 37) {
            double re = 0, im = 0;
            for (int j = 0; j < 512; j++) {
                const double ang = 2 * M_PI * ((i * j) & 511) / 512.0;
                re += z0[j].re * cos(ang) - z0[j].im * sin(ang);
                im += z0[j].re * sin(ang) + z0[j].im * cos(ang);
            }
            worst = fmax(worst, fmax(fabs(re - z[i].re), fabs(im - z[i].im)));
        }
        printf("fft 512: max err %.2e\n", worst);
        CHECK(worst < 1e-3);
        av_fft_end(s);
    }

    /* the codec surface: AAC-LC stereo packets, two contexts fed the same stream agree byte for byte */
    int16_t *pcm[2] = { malloc(HEAAC_MAX_AUDIO_FRAME_SIZE), malloc(HEAAC_MAX_AUDIO_FRAME_SIZE) };
    const int psize = (int)sizeof(HeaacFramePacket) + 2 * 4096;
    uint8_t *pkt = malloc(psize);
    HeaacCodecContext ctx[2];
    for (int k = 0; k < 2; k++) {
        heaac_codec_get_context_defaults(&ctx[k]); ctx[k].sub_id = HEAAC_SUBID_RECORDS(HEAAC_CFG_LC_STEREO);
        CHECK(heaac_codec_open(&ctx[k], &heaac_aac_decoder) == 0);
        CHECK(ctx[k].channels == 2 && ctx[k].frame_size == 1024 && ctx[k].channel_layout == HEAAC_CH_LAYOUT_STEREO);
        CHECK(ctx[k].codec_id == HEAAC_CODEC_ID_AAC && ctx[k].codec_type == 1 && ctx[k].sample_fmt == HEAAC_SAMPLE_FMT_S16);
    }
    unsigned any = 0;
    for (int frame = 0; frame < 3; frame++) {
        HeaacFramePacket hp; memset(&hp, 0, sizeof(hp));
        hp.magic = HEAAC_PACKET_MAGIC; hp.cfg = HEAAC_CFG_LC_STEREO;
        memcpy(pkt, &hp, sizeof(hp));
        float *co = (float *)(pkt + sizeof(hp));
        for (int i = 0; i < 2048; i++) co[i] = frand() * (4096.0f / (1024.0f * 32768.0f));
        for (int k = 0; k < 2; k++) {
            HeaacPacket ap; memset(&ap, 0, sizeof(ap)); ap.data = pkt; ap.size = psize;
            int size = HEAAC_MAX_AUDIO_FRAME_SIZE;
            CHECK(heaac_codec_decode(&ctx[k], pcm[k], &size, &ap) == psize);
            CHECK(ctx[k].frame_number == frame + 1);
            CHECK(size == 1024 * 2 * 2);
        }
        CHECK(!memcmp(pcm[0], pcm[1], 4096));
        for (int i = 0; i < 2048; i++) any |= (unsigned)(pcm[0][i] != 0);
    }
    CHECK(any);
    { /* too small an output buffer, and a packet of the wrong configuration, are refused */
        HeaacPacket ap; memset(&ap, 0, sizeof(ap)); ap.data = pkt; ap.size = psize;
        int size = 100;
        CHECK(heaac_codec_decode(&ctx[0], pcm[0], &size, &ap) < 0);
        ((HeaacFramePacket *)pkt)->cfg = HEAAC_CFG_HEV2; size = HEAAC_MAX_AUDIO_FRAME_SIZE;
        CHECK(heaac_codec_decode(&ctx[0], pcm[0], &size, &ap) < 0);
    }
    for (int k = 0; k < 2; k++) CHECK(heaac_codec_close(&ctx[k]) == 0);
    { /* the reference's own packet in the reference's own records: a context and a packet laid out as libavcodec
         lays them out (transcribed above, filled the way avcodec_alloc_context / av_init_packet fill them) go
         through open / decode / close by pointer cast, as libavcodec's dispatch would pass them.  AAC-LC 48 kHz
         mono from extradata; one SCE with max_sfb = 0 (silence), then END: 32 bits, zero padding behind. */
        static uint8_t asc[2] = { 0x11, 0x88 };
        static uint8_t au[12] = { 0x00, 0xc8, 0x00, 0x07 };
        union ref_AVCodecContext rc; memset(&rc, 0, sizeof(rc));
        rc.h.codec_type = -1;                                    /* AVMEDIA_TYPE_UNKNOWN */
        rc.h.time_base.num = 0; rc.h.time_base.den = 1;
        rc.h.extradata = asc; rc.h.extradata_size = 2;
        HeaacCodecContext *bs = (HeaacCodecContext *)&rc;
        CHECK(heaac_codec_open(bs, &heaac_aac_decoder) == 0);
        CHECK(rc.h.channels == 1 && rc.h.frame_size == 1024 && rc.h.sample_rate == 48000 && rc.h.sample_fmt == 1);
        CHECK(rc.h.codec_id == 0x15002 && rc.h.codec_type == 1 && rc.h.priv_data != NULL);
        CHECK(rc.h.codec == (struct ref_AVCodec *)&heaac_aac_decoder);
        int64_t layout; memcpy(&layout, rc.bytes + REF_AVCTX_CHANNEL_LAYOUT, 8);
        CHECK(layout == 4);                                      /* CH_LAYOUT_MONO */
        for (int frame = 0; frame < 2; frame++) {
            struct ref_AVPacket rp; memset(&rp, 0, sizeof(rp));
            rp.pts = rp.dts = (int64_t)0x8000000000000000ULL;    /* AV_NOPTS_VALUE */
            rp.pos = -1; rp.data = au; rp.size = (int)sizeof(au);
            int size = HEAAC_MAX_AUDIO_FRAME_SIZE;
            CHECK(heaac_codec_decode(bs, pcm[0], &size, (HeaacPacket *)&rp) == (int)sizeof(au));
            CHECK(size == 1024 * 2 && rc.h.frame_number == frame + 1);
            int nz = 0;
            for (int i = 0; i < 1024; i++) nz |= pcm[0][i];
            CHECK(nz == 0);
        }
        /* the callbacks themselves, as the reference's table entry would be called */
        {
            struct ref_AVPacket rp; memset(&rp, 0, sizeof(rp)); rp.data = au; rp.size = (int)sizeof(au);
            int size = HEAAC_MAX_AUDIO_FRAME_SIZE;
            const struct ref_AVCodec *tab = (const struct ref_AVCodec *)&heaac_aac_decoder;
            CHECK(tab->decode(&rc, pcm[0], &size, &rp) == (int)sizeof(au) && size == 2048);
        }
        struct ref_AVPacket bad; memset(&bad, 0, sizeof(bad)); bad.data = asc; bad.size = 2;   /* not an access unit */
        int size = HEAAC_MAX_AUDIO_FRAME_SIZE;
        CHECK(heaac_codec_decode(bs, pcm[0], &size, (HeaacPacket *)&bad) < 0);
        CHECK(heaac_codec_close(bs) == 0 && rc.h.priv_data == NULL && rc.h.codec == NULL);
        /* 960-sample frames (frameLengthFlag) are refused at open, as decode_ga_specific_config refuses them */
        static uint8_t asc960[2] = { 0x11, 0x8c };
        memset(&rc, 0, sizeof(rc)); rc.h.codec_type = -1; rc.h.extradata = asc960; rc.h.extradata_size = 2;
        CHECK(heaac_codec_open(bs, &heaac_aac_decoder) < 0 && rc.h.priv_data == NULL);
    }
    free(pcm[0]); free(pcm[1]); free(pkt);
    return fails;
}

int main(int argc, char **argv)
{
    const char *mode = argc > 1 ? argv[1] : "cpu";
    if (!strcmp(mode, "cpu")) cpu_checks();
    else if (!strcmp(mode, "nodevice")) { cpu_checks(); no_device_checks(); }
    else if (!strcmp(mode, "gpu")) { cpu_checks(); gpu_checks(); }
    else { printf("usage: abi_check cpu|nodevice|gpu\n"); return 2; }
    printf(fails ? "%d check(s) failed\n" : "ok\n", fails);
    return fails ? 1 : 0;
}
