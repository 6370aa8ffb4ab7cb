/* heaac_codec.h -- AVCodec-shaped decoder surface for the HE-AAC DSP path.
 *
 * Mirrors the shape of the reference's codec plugin API so a host decoder can
 * call this library exactly where spectral_to_sample() sits:
 *
 *   struct AVCodec { name, type, id, priv_data_size, init, encode, close,
 *                    decode, ... }            libavcodec/avcodec.h:2675-2711
 *   AVCodec aac_decoder = { "aac", ... aac_decode_init, NULL,
 *                    aac_decode_close, aac_decode_frame }
 *                                             libavcodec/aacdec.c:2128-2142
 *   avcodec_open / avcodec_decode_audio3 / avcodec_close
 *                                             libavcodec/utils.c:462-531, :638-663
 *
 * Two kinds of packet.  By default a packet is what the reference's is: an AAC access unit; decode() parses it
 * on the host (heaac_parse.h) and runs spectral tools + spectral_to_sample() on the GPU.  With
 * avctx->sub_id = HEAAC_SUBID_RECORDS(cfg) a packet is a parser's OUTPUT for one access unit -- dequantised spectrum plus side
 * info -- in the HeaacFramePacket layout below, for hosts that keep their own parser; decode() then does what
 * aac_decode_frame() does from spectral_to_sample() on (aacdec.c:2078-2107).  Either way: float DSP on the
 * GPU, int16 interleaved PCM into the caller's buffer, *data_size = bytes written, return value = bytes
 * consumed (negative on error).
 *
 * One context = one stream = one thread, as in the reference.  Inter-frame
 * state lives in device memory owned by the context.  For throughput use the
 * batched entry points of heaac_dsp.h; this surface exists for drop-in use.
 */
#ifndef HEAAC_CODEC_H
#define HEAAC_CODEC_H

#include <stddef.h>
#include <stdint.h>
#include "heaac_dsp.h"

#ifdef __cplusplus
extern "C" {
#endif

#define HEAAC_MAX_AUDIO_FRAME_SIZE 192000   /* AVCODEC_MAX_AUDIO_FRAME_SIZE, avcodec.h:431 */
#define HEAAC_CODEC_ID_AAC 0x15002          /* CODEC_ID_AAC, avcodec.h */
#define HEAAC_PACKET_MAGIC 0x48454141u      /* "HEAA" */

/* One access unit as the host parsers leave it.  Variable length:
 *   HeaacFramePacket hdr;
 *   float coeffs[ncore][1024];
 *   HeaacSbrFrame sbr;            if cfg is an HE configuration
 *   HeaacPsFrame  ps;             if cfg == HEAAC_CFG_HEV2
 *   HeaacSbrHeader new_header;    if (flags & HEAAC_PKT_NEW_SBR_HEADER); stored
 *                                 at table index sbr.hdr before decoding      */
typedef struct HeaacFramePacket {
    uint32_t magic;
    uint16_t cfg;                 /* HEAAC_CFG_*                             */
    uint16_t flags;
    HeaacIcs ics[2];
} HeaacFramePacket;
#define HEAAC_PKT_NEW_SBR_HEADER 1

/* `AVPacket` of libavcodec 52.78 field for field (avcodec.h:960-1002), so that libavcodec's
 * avcodec_decode_audio3() can hand its own packet to `decode` (utils.c:638-663).  The decoder reads
 * `data` and `size` only, as aac_decode_frame() does (aacdec.c:1976-1977). */
typedef struct HeaacPacket {
    int64_t pts;                  /*  0  avcodec.h:970 */
    int64_t dts;                  /*  8  :976 */
    uint8_t *data;                /* 16  :977 */
    int size;                     /* 24  :978 */
    int stream_index;             /* 28 */
    int flags;                    /* 32 */
    int duration;                 /* 36 */
    void (*destruct)(struct HeaacPacket *);   /* 40 */
    void *priv;                   /* 48 */
    int64_t pos;                  /* 56 */
    int64_t convergence_duration; /* 64  :1001 */
} HeaacPacket;

/* `AVCodecContext` of libavcodec 52.78 on LP64 (avcodec.h:1032-2650, sizeof 1088): the fields the AAC
 * decoder and avcodec_open / _decode_audio3 / _close read or write sit at the reference's offsets under the
 * reference's names (aacdec.c touches extradata, extradata_size, sample_rate, channels, sample_fmt,
 * frame_size, channel_layout, priv_data; utils.c additionally codec, codec_type, codec_id); everything else
 * is opaque padding of the reference's size, never read and never written here.  A context allocated by
 * libavcodec's avcodec_alloc_context() can therefore be passed to these callbacks as it is. */
typedef struct HeaacCodecContext {
    const void *av_class;         /*    0  const AVClass *, avcodec.h:1037 */
    int bit_rate;                 /*    8  :1043 */
    int bit_rate_tolerance;       /*   12  :1051 */
    int flags;                    /*   16  :1058 */
    int sub_id;                   /*   20  :1068  "additional format info": HEAAC_SUBID_RECORDS(cfg) selects
                                   *              parser-output packets (below); anything else = AAC access units */
    int me_method;                /*   24  :1077 */
    uint8_t *extradata;           /*   32  :1090  AudioSpecificConfig (aacdec.c:563-566); NULL: ADTS */
    int extradata_size;           /*   40  :1091 */
    int opaque_44[7];             /*   44  time_base .. rate_emu, :1101-1137 */
    void *draw_horiz_band;        /*   72  :1158 */
    int sample_rate;              /*   80  :1163  output rate (set by init / the first frame) */
    int channels;                 /*   84  :1164  output channels */
    int sample_fmt;               /*   88  :1171  SAMPLE_FMT_S16 (aacdec.c:568) */
    int frame_size;               /*   92  :1177  samples per channel per frame */
    int frame_number;             /*   96  :1178 */
    int opaque_100[13];           /*  100  real_pict_num .. hurry_up, :1180-1245 */
    const struct HeaacCodec *codec;   /* 152  :1247 */
    void *priv_data;              /*  160  :1249 */
    int opaque_168[24];           /*  168  rtp_payload_size .. codec_name, :1251-1291 */
    int codec_type;               /*  264  :1292  AVMEDIA_TYPE_AUDIO */
    int codec_id;                 /*  268  :1293  CODEC_ID_AAC */
    int opaque_272[176];          /*  272  codec_tag .. request_channels and padding, :1308-2470 */
    int64_t channel_layout;       /*  976  :2502  CH_LAYOUT_MONO / _STEREO (output_configure, aacdec.c:224-301) */
    int64_t request_channel_layout;   /* 984  :2509 */
    int opaque_992[24];           /*  992  .. sizeof(AVCodecContext) = 1088 */
} HeaacCodecContext;

#ifdef __cplusplus
#define HEAAC_LAYOUT_ASSERT(c, m) static_assert(c, m)
#else
#define HEAAC_LAYOUT_ASSERT(c, m) _Static_assert(c, m)
#endif
#if defined(__LP64__)
HEAAC_LAYOUT_ASSERT(sizeof(HeaacPacket) == 72 && offsetof(HeaacPacket, data) == 16 && offsetof(HeaacPacket, size) == 24,
                    "HeaacPacket is AVPacket");
HEAAC_LAYOUT_ASSERT(offsetof(HeaacCodecContext, extradata) == 32 && offsetof(HeaacCodecContext, sample_rate) == 80 &&
                    offsetof(HeaacCodecContext, frame_size) == 92 && offsetof(HeaacCodecContext, codec) == 152 &&
                    offsetof(HeaacCodecContext, priv_data) == 160 && offsetof(HeaacCodecContext, codec_id) == 268 &&
                    offsetof(HeaacCodecContext, channel_layout) == 976 && sizeof(HeaacCodecContext) == 1088,
                    "HeaacCodecContext is AVCodecContext");
#endif

/* Packets in the HeaacFramePacket layout (a host parser's output, above) are selected per context the way the
 * reference passes codec-specific format information: through `sub_id`. */
#define HEAAC_SUBID_RECORDS(cfg) (0x48450000 | ((cfg) & 0xff))
#define HEAAC_SUBID_IS_RECORDS(sub_id) (((sub_id) & ~0xff) == 0x48450000)

/* Default (sub_id is not HEAAC_SUBID_RECORDS(..)): packets are AAC access units (raw_data_block, or an ADTS frame), as
 * avcodec_decode_audio3() hands them to aac_decode_frame() (aacdec.c:1973-2107).  The library then parses them
 * itself (heaac_parse.h) and the configuration comes from the stream the way the reference takes it:
 * object type, rate and channel configuration from extradata (decode_audio_specific_config, aacdec.c:432-491)
 * or from the first ADTS header (parse_adts_frame_header, :1935-1971); SBR from the AudioSpecificConfig or,
 * signalled implicitly, from a payload in the FIRST access unit (a first occurrence later is refused,
 * :1666-1669); a mono stream with SBR decodes as Parametric Stereo (:1670-1673, two output channels).
 * Scope of the parser slices: AAC-LC / AAC-Main; one SCE or one CPE per access unit (with Parametric Stereo and coupling
 * channel elements), or the several output elements of channel configuration 3 .. 7 / of a program config element
 * (output_configure, aacdec.c:224-276; SBR per element; no coupling, no Parametric Stereo).  A context keeps up to 63
 * distinct SBR headers of its stream (the derived tables stay on the device); a stream that sends more than that
 * gets -1 from the frame that brings the 64th. */

/* Field for field `struct AVCodec` of libavcodec/avcodec.h:2675-2711 (same order, same types up to
 * the names of the context / packet structs), so that the record can sit in libavcodec's codec list
 * (register_avcodec, utils.c / allcodecs.c:218 REGISTER_ENCDEC) next to the built-in ones. */
typedef struct HeaacCodec {
    const char *name;
    int type;                     /* enum AVMediaType: 1 = AVMEDIA_TYPE_AUDIO */
    int id;                       /* enum CodecID                            */
    int priv_data_size;
    int (*init)(HeaacCodecContext *);
    int (*encode)(HeaacCodecContext *, uint8_t *buf, int buf_size, void *data);
    int (*close)(HeaacCodecContext *);
    int (*decode)(HeaacCodecContext *, void *outdata, int *outdata_size, HeaacPacket *avpkt);
    int capabilities;             /* CODEC_CAP_*: none                       */
    struct HeaacCodec *next;
    void (*flush)(HeaacCodecContext *);
    const void *supported_framerates;   /* const AVRational *: video only    */
    const int *pix_fmts;                /* const enum PixelFormat *: video   */
    const char *long_name;
    const int *supported_samplerates;
    const int *sample_fmts;             /* const enum SampleFormat *: {SAMPLE_FMT_S16, SAMPLE_FMT_NONE} */
    const int64_t *channel_layouts;     /* {CH_LAYOUT_MONO, CH_LAYOUT_STEREO, 0} */
} HeaacCodec;
#define HEAAC_MEDIA_TYPE_UNKNOWN (-1)   /* AVMEDIA_TYPE_UNKNOWN, avcodec.h */
#define HEAAC_SAMPLE_FMT_NONE (-1)      /* avcodec.h:376-378 */
#define HEAAC_SAMPLE_FMT_S16  1
#define HEAAC_CH_LAYOUT_MONO   0x4      /* CH_FRONT_CENTER, avcodec.h:388, 413 */
#define HEAAC_CH_LAYOUT_STEREO 0x3      /* CH_FRONT_LEFT | CH_FRONT_RIGHT, :386-387, 414 */

extern HeaacCodec heaac_aac_decoder;

/* avcodec_get_context_defaults2(s, AVMEDIA_TYPE_UNKNOWN) (options.c), for callers that do not get their
 * context from libavcodec's avcodec_alloc_context(): zeroes the record, codec_type = AVMEDIA_TYPE_UNKNOWN. */
void heaac_codec_get_context_defaults(HeaacCodecContext *avctx);
/* avcodec_open (utils.c:462-531): allocates priv_data, adopts or checks codec_type / codec_id, calls codec->init. */
int heaac_codec_open(HeaacCodecContext *avctx, HeaacCodec *codec);
/* avcodec_decode_audio3 (utils.c:638-663): checks *frame_size_ptr >=
 * HEAAC_MAX_AUDIO_FRAME_SIZE, calls codec->decode. */
int heaac_codec_decode(HeaacCodecContext *avctx, int16_t *samples, int *frame_size_ptr, HeaacPacket *avpkt);
/* avcodec_close (utils.c:533-560) */
int heaac_codec_close(HeaacCodecContext *avctx);

#ifdef __cplusplus
}
#endif
#endif /* HEAAC_CODEC_H */
