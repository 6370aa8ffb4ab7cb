/* heaac_parse.h -- host-side AAC bitstream parser feeding the batched GPU path.
 *
 * First slice: AAC-LC / AAC-Main access units made of ONE single_channel_element or ONE
 * channel_pair_element (channel configurations 1 and 2), with data_stream and fill elements skipped.
 * What it replaces in the reference (libavcodec):
 *
 *   heaac_asc_parse          ff_mpeg4audio_get_config            mpeg4audio.c:79-143
 *   heaac_adts_parse_header  ff_aac_parse_header (ADTS fixed+variable header)   aac_parser.c:29-70
 *   heaac_aac_parse_frame    aac_decode_frame's element loop     aacdec.c:1973-2075
 *                            decode_ics / decode_cpe             :1334-1388, :1453-1492
 *                            decode_ics_info, decode_prediction  :622-742
 *                            decode_band_types, decode_scalefactors, decode_pulses, decode_tns,
 *                            decode_mid_side_stereo              :755-945
 *                            decode_spectrum_and_dequant         :988-1245  (noise bands left to the GPU)
 *   heaac_aac_parse_batch    the same over many independent streams on host threads
 *
 * Output = exactly what heaac_spectral_tools_batch + heaac_lc_decode_batch (heaac_dsp.h) take: the
 * dequantised, scaled spectrum, the window info of this and the previous frame, and the side info of the
 * spectral tools.  The arithmetic of the dequantisation is the reference's (same products, same order).
 * SBR / PS payloads (fill elements of type EXT_SBR_DATA) are located but not parsed in this slice:
 * their position is reported so that a later SBR parser can take them.
 */
#ifndef HEAAC_PARSE_H
#define HEAAC_PARSE_H

#include <stddef.h>
#include <stdint.h>
#include "heaac_dsp.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
    HEAAC_PARSE_OK          =  0,
    HEAAC_PARSE_ERR_DATA    = -1,   /* invalid or reserved value in the bitstream (the reference returns -1) */
    HEAAC_PARSE_ERR_OVERREAD = -2,  /* ran past the end of the access unit */
    HEAAC_PARSE_ERR_UNSUPPORTED = -3, /* valid AAC outside this slice: CCE, PCE, LTP, SSR gain control,
                                         more than one SCE / CPE per access unit */
    HEAAC_PARSE_ERR_ARG     = -4,
};

enum { HEAAC_AOT_AAC_MAIN = 1, HEAAC_AOT_AAC_LC = 2, HEAAC_AOT_SBR = 5, HEAAC_AOT_PS = 29 };

/* MPEG4AudioConfig (mpeg4audio.h:28-38) */
typedef struct HeaacAacConfig {
    int object_type;
    int sampling_index;
    int sample_rate;
    int chan_config;
    int sbr;                      /* -1 implicit, 1 presence flag */
    int ext_object_type;
    int ext_sampling_index;
    int ext_sample_rate;
    int ext_chan_config;
    int ps;                       /* -1 implicit, 1 presence flag */
} HeaacAacConfig;

/* AudioSpecificConfig -> config.  Returns the bit offset of the specific config (as the reference
 * does) or a negative HEAAC_PARSE_ERR_*. */
int heaac_asc_parse(HeaacAacConfig *c, const uint8_t *buf, int size);

/* AACADTSHeaderInfo (aac_parser.h / aac_parser.c:29-70) */
typedef struct HeaacAdtsHeader {
    int sample_rate;
    int samples;                  /* 1024 x raw data blocks */
    int bit_rate;
    int object_type;
    int sampling_index;
    int chan_config;
    int crc_absent;
    int num_aac_frames;
    int frame_length;             /* bytes, header included */
} HeaacAdtsHeader;

/* Returns the header size in bytes (7 or 9) or a negative error:
 * -1 no sync word, -2 reserved sampling index, -3 frame length shorter than the header. */
int heaac_adts_parse_header(HeaacAdtsHeader *h, const uint8_t *buf, int size);

/* What the parser carries from frame to frame of one stream (IndividualChannelStream
 * window_sequence[1] / use_kb_window[1], aac.h:137-138).  Zero-initialise for a new stream. */
typedef struct HeaacAacStream {
    uint8_t window_sequence[2];   /* of the previous frame, per channel */
    uint8_t use_kb_window[2];
    uint8_t pad[4];
} HeaacAacStream;

typedef struct HeaacAacFrameInfo {
    int channels;                 /* 1 (SCE) or 2 (CPE) */
    int bits_consumed;
    int sbr_payload_bit;          /* bit offset of an EXT_SBR_DATA(_CRC) fill payload after its 4-bit type, -1: none */
    int sbr_payload_bytes;        /* its length in bytes (the `cnt` of decode_extension_payload) */
    int sbr_crc;
} HeaacAacFrameInfo;

/* One access unit (raw_data_block; an ADTS header in front is skipped as aac_decode_frame does).
 *   coeffs [2][1024]  dequantised spectrum per channel (channel 1 untouched for an SCE); NOISE_BT bands
 *                     are zero here: heaac_spectral_tools_batch fills them from the stream's generator
 *   ics [2]           window info (this frame [0], previous frame [1])
 *   tools             side info of the spectral tools (M/S, intensity, TNS, PNS, prediction)
 * Returns HEAAC_PARSE_OK or a negative error; on error the stream state is left as it was. */
int heaac_aac_parse_frame(const HeaacAacConfig *cfg, HeaacAacStream *st,
                          const uint8_t *au, int size,
                          float *coeffs, HeaacIcs *ics, HeaacToolsFrame *tools,
                          HeaacAacFrameInfo *info);

/* n independent streams, one access unit each, on `threads` host threads (<= 0: one per online CPU).
 *   au[n], size[n]         access units
 *   coeffs [n][2][1024], ics [n][2], tools [n], info [n] (may be NULL), status [n] per-frame result
 * Returns the number of frames that failed (0 = all parsed). */
int heaac_aac_parse_batch(const HeaacAacConfig *cfg, HeaacAacStream *st,
                          const uint8_t *const *au, const int *size, size_t n,
                          float *coeffs, HeaacIcs *ics, HeaacToolsFrame *tools,
                          HeaacAacFrameInfo *info, int *status, int threads);

/* SHA-256-free integrity hook for tests: FNV-1a of the generated ISO tables (codes, lengths, band offsets). */
uint64_t heaac_aac_tables_fingerprint(void);

#ifdef __cplusplus
}
#endif
#endif /* HEAAC_PARSE_H */
