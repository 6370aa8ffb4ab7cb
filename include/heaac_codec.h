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
 * Two kinds of packet.  With cfg = HEAAC_CFG_FROM_STREAM a packet is what the reference's is: an AAC access
 * unit; decode() parses it on the host (heaac_parse.h) and runs spectral tools + spectral_to_sample() on the GPU.
 * With an explicit cfg a packet is a parser's OUTPUT for one access unit -- dequantised spectrum plus side
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

/* AVPacket subset (avcodec.h:1033-1080) */
typedef struct HeaacPacket {
    const uint8_t *data;
    int size;
} HeaacPacket;

/* AVCodecContext subset (avcodec.h:1113-2650): what the AAC decoder reads/sets */
typedef struct HeaacCodecContext {
    int sample_rate;              /* output rate (set by init from cfg)      */
    int channels;                 /* output channels                         */
    int frame_size;               /* samples per channel per frame           */
    int cfg;                      /* HEAAC_CFG_* chosen by the caller, or HEAAC_CFG_FROM_STREAM */
    const struct HeaacCodec *codec;
    void *priv_data;
    const uint8_t *extradata;     /* AudioSpecificConfig (avctx->extradata, aacdec.c:563-566); NULL: ADTS */
    int extradata_size;
} HeaacCodecContext;

/* cfg = HEAAC_CFG_FROM_STREAM: packets are AAC access units (raw_data_block, or an ADTS frame), as
 * avcodec_decode_audio3() hands them to aac_decode_frame() (aacdec.c:1973-2107).  The library then parses them
 * itself (heaac_parse.h) and the configuration comes from the stream the way the reference takes it:
 * object type, rate and channel configuration from extradata (decode_audio_specific_config, aacdec.c:432-491)
 * or from the first ADTS header (parse_adts_frame_header, :1935-1971); SBR from the AudioSpecificConfig or,
 * signalled implicitly, from a payload in the FIRST access unit (a first occurrence later is refused,
 * :1666-1669); a mono stream with SBR decodes as Parametric Stereo (:1670-1673, two output channels).
 * Scope of the parser slices: one SCE or one CPE per access unit, AAC-LC / AAC-Main.  A context keeps up to 63
 * distinct SBR headers of its stream (the derived tables stay on the device); a stream that sends more than that
 * gets -1 from the frame that brings the 64th. */
#define HEAAC_CFG_FROM_STREAM (-1)

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
#define HEAAC_SAMPLE_FMT_NONE (-1)      /* avcodec.h:376-378 */
#define HEAAC_SAMPLE_FMT_S16  1
#define HEAAC_CH_LAYOUT_MONO   0x4      /* CH_FRONT_CENTER, avcodec.h:388, 413 */
#define HEAAC_CH_LAYOUT_STEREO 0x3      /* CH_FRONT_LEFT | CH_FRONT_RIGHT, :386-387, 414 */

extern HeaacCodec heaac_aac_decoder;

/* avcodec_open (utils.c:462-531): allocates priv_data, calls codec->init. */
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
