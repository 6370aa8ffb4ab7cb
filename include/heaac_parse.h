/* heaac_parse.h -- host-side AAC bitstream parser feeding the batched GPU path.
 *
 * First slice: AAC-LC / AAC-Main access units made of ONE single_channel_element or ONE
 * channel_pair_element (channel configurations 1 and 2), with data_stream and fill elements skipped; the layout
 * entries further down take the access units of channel configurations 3 .. 7 and of program config elements.
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
 * SBR / PS payloads (fill elements of type EXT_SBR_DATA) are located by this slice and parsed by the
 * second one (heaac_sbr_parse_payload, below).
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
    HEAAC_PARSE_ERR_UNSUPPORTED = -3, /* valid AAC outside the entry it was handed to: a coupling element or a second
                                         SCE / CPE in heaac_aac_parse_frame, ...; and what the reference itself reports as a
                                         missing feature: LTP (aacdec.c:694), SSR gain control (:1373), 960-sample frames (:409) */
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
/* GASpecificConfig at the bit offset heaac_asc_parse returned (decode_ga_specific_config,
 * aacdec.c:401-452): 0, or HEAAC_PARSE_ERR_UNSUPPORTED for 960-sample frames (frameLengthFlag) and for
 * channel configuration 0 (program config element), HEAAC_PARSE_ERR_OVERREAD when the buffer ends first. */
int heaac_ga_specific_config(const HeaacAacConfig *c, const uint8_t *buf, int size, int bit_offset);

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

/* ---- a raw ADTS (.aac) buffer -> access units (libavformat/raw.c:666-717, libavcodec/aac_ac3_parser.c:26-100) ---- */
enum {
    HEAAC_ADTS_FRAME = 0,         /* one ADTS frame, header included: an access unit for heaac_aac_parse_frame /
                                   * heaac_heaac_parse_frame / heaac_codec_decode */
    HEAAC_ADTS_JUNK = 1,          /* bytes no header claims (damage, padding); the reference's parser passes such
                                   * spans to the decoder as packets, which refuses them */
    HEAAC_ADTS_TRUNCATED = 2,     /* a header whose frame runs past the end of the buffer */
    HEAAC_ADTS_TAG = 3            /* an ID3v2 tag at the start (stepped over by the demuxer, raw.c:677-679, :711) */
};
typedef struct HeaacAdtsPacket {
    size_t offset, size;          /* span of the buffer */
    int kind;                     /* HEAAC_ADTS_* */
    int header_size;              /* frames: 7, or 9 with a CRC */
} HeaacAdtsPacket;

/* adts_aac_probe (raw.c:666-702): 51 if the buffer starts (behind an ID3v2 tag) with three or more frames in
 * step, 50 for a run of more than 500 anywhere, 25 for a run of three, 1 for a single header, else 0. */
int heaac_adts_probe(const uint8_t *buf, size_t size);

/* Walk a whole .aac buffer into packets, frames behind each other as ff_aac_ac3_parse delivers them.  Directly
 * behind a good frame a header is taken as the reference takes it (sync word, legal sampling index, length >= 7);
 * a header found while SEARCHING (start of the buffer, behind damage) must be followed by another header one frame
 * length on (the probe's rule; the reference's streaming parser cannot look ahead and takes the first seven bytes
 * that parse).  Writes up to max_out packets (out may be NULL to count) and returns how many the buffer holds, or
 * a negative HEAAC_PARSE_ERR_*.  *first_header (may be NULL) receives the first frame's header: the stream's
 * configuration as parse_adts_frame_header (aacdec.c:1935-1971) would take it. */
long heaac_adts_split(const uint8_t *buf, size_t size, HeaacAdtsPacket *out, size_t max_out, HeaacAdtsHeader *first_header);

/* What the parser carries from frame to frame of one stream (IndividualChannelStream
 * window_sequence[1] / use_kb_window[1], aac.h:137-138).  Zero-initialise for a new stream. */
typedef struct HeaacAacStream {
    uint8_t window_sequence[2];   /* of the previous frame, per channel */
    uint8_t use_kb_window[2];
    uint8_t cce_window_sequence[16];  /* the same for the coupling channels, by instance tag (che[TYPE_CCE][tag]) */
    uint8_t cce_use_kb_window[16];
    uint8_t mapped_tag;           /* heaac_aac_parse_frame(_ex): 1 + the instance tag the stream's output element was first
                                     seen with, 0 before.  get_che maps the one element of a channel configuration 1 / 2
                                     stream to the first tag it meets and to no other (ac->tag_che_map, tags_mapped,
                                     aacdec.c:131-177): a later unit whose element carries another tag is refused
                                     ("channel element %d.%d is not allocated", :2011-2015) */
    uint8_t pad[3];
} HeaacAacStream;

typedef struct HeaacAacFrameInfo {
    int channels;                 /* 1 (SCE) or 2 (CPE) */
    int bits_consumed;
    int sbr_payload_bit;          /* bit offset of an EXT_SBR_DATA(_CRC) fill payload after its 4-bit type, -1: none */
    int sbr_payload_bytes;        /* its length in bytes (the `cnt` of decode_extension_payload) */
    int sbr_crc;
    int elem_id;                  /* instance tag of the output element */
    int n_cce;                    /* coupling elements found (heaac_aac_parse_frame_ex, heaac_aac_parse_frame_layout_ex) */
    int sbr_misplaced;            /* 1: the payload's fill element does not stand directly behind the SCE / CPE: the
                                     reference hands its SBR reader the type of the element in between, which reads a
                                     header if there is one and then switches the element's SBR off (aacdec.c:2059,
                                     aacsbr.c:996-1000).  Pass HEAAC_SBR_MISPLACED to heaac_sbr_parse_payload. */
    int refused;                  /* after a negative return of heaac_aac_parse_frame(_ex) / heaac_heaac_parse_frame(_ex):
                                     HEAAC_REFUSED_* (the other fields are then 0, sbr_payload_bit -1) */
} HeaacAacFrameInfo;

/* A refused access unit gives no samples, but the reference does not undo what its element decoders had done by
 * then (aac_decode_frame leaves its element loop at the first error, aacdec.c:2069-2070):
 *   - decode_ics_info has moved the channel's window history on to the refused unit's, or cleared it where the
 *     refusal is its own (memset of the IndividualChannelStream, :650, 687-705);
 *   - decode_spectrum_and_dequant has drawn the numbers of the noise bands it passed (:1016-1029);
 *   - apply_prediction has stepped the AAC-Main predictors of a channel that was completed (:1381, 1486-1489).
 * HEAAC_REFUSED_AS_REFERENCE: the refusal is one the reference makes at the same bit of the unit (a reserved value,
 * a range check, one of its three end-of-unit checks) and `st` now holds the window history its decoder is left
 * with.  Clear for the refusals that are this parser's alone -- a read past the end of the unit where the reference
 * reads on unchecked, element combinations it is not built for, any unit with a coupling element: `st` is then
 * untouched, as if the unit had not been there.
 * HEAAC_REFUSED_RUN_TOOLS (only with the former): `tools` and `coeffs` have been rewritten into records under which
 * heaac_spectral_tools_batch draws from the stream's noise generator and steps its predictors exactly as far as the
 * reference had; run it on them as for a good unit (both channels of a pair) and discard the coefficients.
 * heaac_aac_parse_frame_layout(_ex) does the same per element: `st[e]` of every element the unit got through, and of
 * the one the refusal stands in, holds the reference's window history; with HEAAC_REFUSED_RUN_TOOLS the elements
 * marked `present` in `elem` (the completed ones with their own records, the refused one rewritten) are to go through
 * the spectral tools in `seq` order on the stream's one noise generator.
 * The SBR / PS side of a refused unit is not followed up: an extension payload in front of the refusal, which the
 * reference has read by then, is not read here. */
#define HEAAC_REFUSED_AS_REFERENCE 1
#define HEAAC_REFUSED_RUN_TOOLS    2

/* What heaac_aac_parse_frame_ex adds for access units that carry coupling channel elements: per slot (ascending
 * instance tag, HEAAC_MAX_CCE slots) the element's coupling record with its gain lists resolved against the target
 * element (decode_cce aacdec.c:1503-1570 + the index walk of apply_channel_coupling :1870-1898), its own spectrum
 * and the side info of ITS spectral tools (a coupling channel is an individual channel stream: noise substitution,
 * prediction and TNS apply to it before it couples, spectral_to_sample :1907-1916). */
typedef struct HeaacCceOut {
    HeaacCceFrame *cce;           /* [HEAAC_MAX_CCE] */
    float *coeffs;                /* [HEAAC_MAX_CCE][1024] */
    HeaacIcs *ics;                /* [HEAAC_MAX_CCE] window info of the coupling channels (AFTER_IMDCT ones are transformed) */
    HeaacToolsFrame *tools;       /* [HEAAC_MAX_CCE]: channel 0 = the coupling channel */
    struct HeaacAacElementInfo *elem;   /* [HEAAC_MAX_CCE] or NULL (heaac_aac_parse_frame_layout_ex only): where each coupling
                                     element stands and the SBR payload behind it -- a coupling channel that couples
                                     AFTER_IMDCT goes through ff_sbr_apply like an SCE (aacdec.c:1920-1927).  NULL: such
                                     a payload is HEAAC_PARSE_ERR_UNSUPPORTED. */
} HeaacCceOut;

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

/* The same for an access unit with one SCE or one CPE as the output element: data stream and fill elements
 * anywhere, program config elements (read past: decode_pce :303-357; the channel layout stays the one of the
 * configuration), and up to HEAAC_MAX_CCE coupling channel elements into `cce` (NULL: an access unit with a
 * coupling element is HEAAC_PARSE_ERR_UNSUPPORTED, as in heaac_aac_parse_frame), their gain lists resolved against
 * the output element as (type of the configuration, tag 0).  A second SCE / CPE / LFE, a third coupling element, or
 * more than HEAAC_MAX_CCE_LINKS gain lists of one element landing on the target are HEAAC_PARSE_ERR_UNSUPPORTED.
 * This is the record-level entry (it feeds heaac_spectral_tools_batch_ex and the coupling tests).  aac_decode_frame
 * itself only knows the coupling elements a program config element has named (che_configure :198-212; get_che
 * :132-177 finds no other, "channel element 2.%d is not allocated"): through the codec surface such access units
 * decode in a channel-configuration-0 stream, by heaac_aac_parse_frame_layout_ex below, and fail in any other.
 * coeff_channels: 1 or 2 -- the channel stride of `coeffs`, `ics`: 2 = [2][1024] / [2] as above, 1 = [1][1024] /
 * [1] packed for mono streams (a CPE then fails with HEAAC_PARSE_ERR_ARG). */
int heaac_aac_parse_frame_ex(const HeaacAacConfig *cfg, HeaacAacStream *st,
                             const uint8_t *au, int size, int coeff_channels,
                             float *coeffs, HeaacIcs *ics, HeaacToolsFrame *tools,
                             const HeaacCceOut *cce, HeaacAacFrameInfo *info);

/* ---- channel layouts: several output elements per access unit (widened) ------------------
 *   heaac_aac_layout_default      set_default_channel_config + output_configure     aacdec.c:359-398, :224-276
 *   heaac_aac_layout_from_pce     decode_pce + output_configure (channel_config 0)  :303-357
 *   heaac_aac_parse_frame_layout  aac_decode_frame's element loop with get_che      :113-183, :1999-2075
 * A layout is the list of the output elements (SCE / CPE / LFE) in the order their channels leave the decoder
 * (che_configure: `output_data[channels++]`), plus what get_che needs to find an element of the bitstream in it:
 * for the channel configurations 1..7 the elements are taken BY POSITION -- the n-th output element of a stream must
 * be of the type the configuration has there, whatever its instance tag, and keeps that tag from then on -- for a
 * program config element by (type, tag).  Coupling channel elements are no output elements: a program config
 * element's are kept in slot_of / tag_map[HEAAC_ELEM_CCE][tag] as 1 + their place in ascending tag order (the order
 * apply_channel_coupling walks them in, :1876); a channel configuration 1..7 has none. */
enum { HEAAC_ELEM_SCE = 0, HEAAC_ELEM_CPE = 1, HEAAC_ELEM_CCE = 2, HEAAC_ELEM_LFE = 3 };
#define HEAAC_MAX_ELEMENTS 16     /* output elements of one layout (channel configuration 7 has five) */
#define HEAAC_MAX_LAYOUT_CHANNELS 16

typedef struct HeaacAacElementSlot {
    uint8_t type;                 /* HEAAC_ELEM_SCE / _CPE / _LFE */
    uint8_t id;                   /* which element of that type (the index into the reference's che[type][]) */
    uint8_t channels;             /* 1 or 2 */
    uint8_t first_channel;        /* position of its first channel in the interleaved output */
} HeaacAacElementSlot;

typedef struct HeaacAacLayout {
    int32_t chan_config;          /* 1..7, or 0: from a program config element */
    int32_t n_elements;
    int32_t channels;             /* avctx->channels */
    int32_t tags_mapped;          /* get_che: output elements of the stream met so far (configurations 1..7) */
    int64_t channel_layout;       /* avctx->channel_layout: aac_channel_layout[] (aacdectab.h:84-93), 0 for a PCE */
    HeaacAacElementSlot elem[HEAAC_MAX_ELEMENTS];   /* in output order */
    int8_t  slot_of[4][16];       /* (type, id) -> index into elem[] + 1; 0: not in the layout */
    int8_t  tag_map[4][16];       /* get_che's tag_che_map: (type, instance tag) -> index into elem[] + 1; 0: not met yet */
} HeaacAacLayout;

/* The layout of channel configuration 1..7 (SCE = centre, CPE 0 = left / right ... in the reference's output order:
 * 5.1 leaves as L R C LFE Ls Rs).  Returns 0, or HEAAC_PARSE_ERR_DATA for another value. */
int heaac_aac_layout_default(HeaacAacLayout *l, int chan_config);

/* The layout a program_config_element describes: `bit_offset` = the element's first bit behind its 4-bit instance
 * tag (in an access unit: behind the 3-bit element type and the tag; in a GASpecificConfig: where
 * heaac_ga_specific_config found channel configuration 0 -- use heaac_asc_layout).  Output order as
 * output_configure builds it: ids ascending, for every id SCE, CPE, then LFE.  *bits_used (may be NULL) receives the
 * element's length.  HEAAC_PARSE_ERR_OVERREAD when the buffer ends inside it, HEAAC_PARSE_ERR_UNSUPPORTED for more
 * than HEAAC_MAX_LAYOUT_CHANNELS channels. */
int heaac_aac_layout_from_pce(HeaacAacLayout *l, const uint8_t *buf, int size, int bit_offset, int *bits_used);

/* The layout of a stream that configures itself: the program config element an access unit of a channel-
 * configuration-0 ADTS stream carries ahead of its channel elements (aac_decode_frame :2036-2046, OC_TRIAL_PCE).  `au`
 * with or without its ADTS header; data stream and fill elements may stand in front of the program config element.
 * HEAAC_PARSE_ERR_DATA when a channel element (nothing is allocated yet) or an SBR payload comes first, or there is none. */
int heaac_aac_layout_from_au(HeaacAacLayout *l, const uint8_t *au, int size);

/* AudioSpecificConfig -> configuration and layout in one step (decode_audio_specific_config, aacdec.c:462-493):
 * heaac_asc_parse, the GASpecificConfig checks, and the layout of its channel configuration or of the program
 * config element it carries.  Returns 0 or a negative HEAAC_PARSE_ERR_*. */
int heaac_asc_layout(HeaacAacConfig *c, HeaacAacLayout *l, const uint8_t *buf, int size);

/* One output element of an access unit, as heaac_aac_parse_frame_layout found it */
typedef struct HeaacAacElementInfo {
    uint8_t present;              /* the access unit carried this element */
    uint8_t type;                 /* its type in the bitstream (an SCE may stand where a 5.1 / 7.1 layout has its LFE, :146-152) */
    uint8_t tag;                  /* its instance tag in the bitstream */
    uint8_t seq;                  /* position among the access unit's output elements (bitstream order: the order the
                                   * noise generator of heaac_spectral_tools_batch has to run through them) */
    uint8_t sbr_crc;
    uint8_t sbr_misplaced;        /* as in HeaacAacFrameInfo; also 1 behind an LFE, whose type the SBR reader refuses
                                   * the same way (aacsbr.c:986-1000) */
    uint8_t pad[2];
    int32_t sbr_payload_bit;      /* as in HeaacAacFrameInfo: the first EXT_SBR_DATA fill payload behind the element
                                   * (before the next channel element), -1: none */
    int32_t sbr_payload_bytes;
} HeaacAacElementInfo;

/* One access unit of a multi-element stream.  Everything is per slot of the layout (index into layout->elem[]):
 *   st [n_elements]              window history per element (zero-initialise for a new stream)
 *   coeffs [n_elements][2][1024], ics [n_elements][2], tools [n_elements], elem [n_elements]
 * The layout's tag map is updated as get_che updates tag_che_map.  An in-band program config element is read
 * past (the reference ignores it too once the layout is settled, :2041-2043).  info->channels = the layout's.
 * Returns HEAAC_PARSE_OK; HEAAC_PARSE_ERR_DATA for an element the layout has no place for ("channel element is
 * not allocated", :2006-2010) -- a coupling channel element the layout does not name included;
 * HEAAC_PARSE_ERR_UNSUPPORTED for a coupling channel element it does name (heaac_aac_parse_frame_layout_ex takes
 * those) and for an SBR payload that does not directly follow its element (the reference hands such a payload to its SBR reader with the type of
 * the data / fill element in between, which switches that element's SBR off); other errors as heaac_aac_parse_frame.  On error the stream states are left as they were (the tag map keeps what it learned, as
 * the reference's does). */
int heaac_aac_parse_frame_layout(const HeaacAacConfig *cfg, HeaacAacLayout *layout, HeaacAacStream *st,
                                 const uint8_t *au, int size,
                                 float *coeffs, HeaacIcs *ics, HeaacToolsFrame *tools,
                                 HeaacAacElementInfo *elem, HeaacAacFrameInfo *info);

/* The same with the coupling channel elements of the layout (aac_decode_frame :2025-2027 decode_cce):
 *   cce->cce    [n_elements][HEAAC_MAX_CCE]  per OUTPUT slot the coupling elements with the gain lists that land on
 *                                            that element (apply_channel_coupling compares the target list with the
 *                                            element's type and its place in ac->che[type][], :1880-1881 -- for a
 *                                            program-config layout its tag); the index is the coupling element's
 *                                            place in the layout (ascending tag), `present` = 0 where the access
 *                                            unit left it out
 *   cce->coeffs [HEAAC_MAX_CCE][1024], cce->ics [HEAAC_MAX_CCE], cce->tools [HEAAC_MAX_CCE] (channel 0)
 * The coupling channels' window history is st[0].cce_window_sequence / cce_use_kb_window.  info->n_cce = coupling
 * elements found (HEAAC_MAX_CCE = 16: one slot per instance tag, every coupling element a layout can name).  A
 * coupling element with more than
 * HEAAC_MAX_CCE_LINKS gain lists on one output element: HEAAC_PARSE_ERR_UNSUPPORTED. */
int heaac_aac_parse_frame_layout_ex(const HeaacAacConfig *cfg, HeaacAacLayout *layout, HeaacAacStream *st,
                                    const uint8_t *au, int size,
                                    float *coeffs, HeaacIcs *ics, HeaacToolsFrame *tools,
                                    HeaacAacElementInfo *elem, const HeaacCceOut *cce, HeaacAacFrameInfo *info);

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

/* ------------------------------------------------------------------------------------------------ */
/* Second slice: the SBR extension payload and the Parametric Stereo data inside it                   */
/* ------------------------------------------------------------------------------------------------ */
/*   heaac_sbr_parse_payload  ff_decode_sbr_extension             aacsbr.c:1044-1090
 *                            read_sbr_header                     :207-262
 *                            sbr_reset -> heaac_sbr_make_header  :1022-1033 (tables: sbr_header.c)
 *                            read_sbr_data, read_sbr_single_channel_element,
 *                            read_sbr_channel_pair_element       :928-1020
 *                            read_sbr_grid, copy_sbr_grid, read_sbr_dtdf, read_sbr_invf,
 *                            read_sbr_envelope, read_sbr_noise   :609-898
 *                            read_sbr_extension                  :900-926
 *                            ff_ps_read_data, read_iid/icc/ipdopd_data, ps_read_extension_data
 *                                                                aacps.c:84-279
 * Output = the records heaac_he_decode_batch takes (heaac_dsp.h): one HeaacSbrFrame (+ HeaacPsFrame) per
 * access unit, frames referring by index to a table of derived header records.
 *
 * Where this parser is stricter than the reference (each case is malformed input on which the reference
 * goes on with undefined or non-finite values; here the element is dropped the way the reference drops one
 * after a grid error -- start = 0, "pure upsampling" -- and the call returns HEAAC_PARSE_ERR_DATA):
 *   - two SBR time borders coincide (the reference accepts, then divides by zero in sbr_env_estimate);
 *   - an accumulated envelope / noise scalefactor leaves 0..255 (the record's uint8);
 *   - explicit PS borders that do not ascend, or an IID envelope carried over from a frame of the finer
 *     quantiser into one of the coarser (the reference indexes its tables with it) (PS only: ps.start = 0);
 *   - after a failed element the channel state is rolled back to the previous frame's (the reference keeps
 *     a half-written one, bs_num_env = 8 included, and indexes with it on the next frame);
 *   - a header whose limiter table comes out empty (a short SBR range over a dropped patch; the reference then
 *     applies gains left by earlier frames) is treated like a failed sbr_reset;
 *   - after a failed sbr_reset the next header resets again (the reference would accept it as "unchanged"
 *     and read data against half-built tables); the very first header of a stream always resets.
 * An access unit WITHOUT an SBR payload (the reference re-applies the previous frame's already dequantised
 * envelopes, i.e. garbage) is reported as HEAAC_PARSE_NO_SBR with a start = 0 record. */

#define HEAAC_PARSE_NO_SBR 1      /* informational: no SBR payload in this access unit */

/* Table of derived SBR headers shared by all streams of a batch: entry 0 is the null header (kx = 32,
 * m = 0: the decoder before any header, aacsbr.c:130); identical headers share one entry.  Thread-safe;
 * the storage never moves, so heaac_sbr_table_data() stays valid while entries are added. */
typedef struct HeaacSbrHeaderTable HeaacSbrHeaderTable;
HeaacSbrHeaderTable *heaac_sbr_table_create(size_t capacity);          /* capacity <= 65535 */
void   heaac_sbr_table_destroy(HeaacSbrHeaderTable *t);
size_t heaac_sbr_table_count(const HeaacSbrHeaderTable *t);
const HeaacSbrHeader *heaac_sbr_table_data(const HeaacSbrHeaderTable *t);

/* What SBRData (sbr.h:62-105) carries from frame to frame */
typedef struct HeaacSbrChanState {
    uint8_t bs_num_env;           /* of the last parsed frame; 0 before any */
    uint8_t bs_num_noise;
    uint8_t bs_amp_res;
    uint8_t bs_frame_class;
    int8_t  e_a[2];               /* e_a[1] = -1 in a new stream */
    uint8_t bs_add_harmonic_flag;
    uint8_t t_env_num_env_old;
    uint8_t bs_freq_res[8];
    uint8_t t_env[8];
    uint8_t t_q[3];
    uint8_t bs_df_env[5];
    uint8_t bs_df_noise[2];
    uint8_t bs_invf_mode[2][5];
    uint8_t bs_add_harmonic[48];
    int32_t env_facs[6][48];      /* accumulated integers; row 0 = the last envelope of the previous frame */
    int32_t noise_facs[3][5];
} HeaacSbrChanState;

/* PSContext (aacps.h:41-61) bitstream side */
typedef struct HeaacPsState {
    uint8_t start, enable_iid, iid_quant, nr_iid_par, nr_ipdopd_par, enable_icc, icc_mode, nr_icc_par;
    uint8_t enable_ext, frame_class, num_env_old, num_env, enable_ipdopd, is34bands, is34bands_old, pad;
    int8_t  border_position[8];
    int8_t  iid_par[5][34], icc_par[5][34], ipd_par[5][34], opd_par[5][34];
} HeaacPsState;

/* SpectralBandReplication (sbr.h:112-160) bitstream side.  heaac_sbr_stream_init() for a new stream. */
typedef struct HeaacSbrStream {
    uint8_t start, reset, have_spectrum, bs_coupling;
    uint8_t bs_start_freq, bs_stop_freq, bs_xover_band, bs_freq_scale, bs_alter_scale, bs_noise_bands;
    uint8_t bs_amp_res_header, bs_limiter_bands, bs_limiter_gains, bs_interpol_freq, bs_smoothing_mode;
    uint8_t pad;
    uint8_t kx[2], m[2];
    uint32_t hdr;                 /* index of the current header in the table (0 = none yet) */
    HeaacSbrChanState data[2];
    HeaacPsState ps;
} HeaacSbrStream;

void   heaac_sbr_stream_init(HeaacSbrStream *st, size_t n);
size_t heaac_sbr_stream_bytes(void);

enum { HEAAC_SBR_ALLOW_PS = 1, HEAAC_SBR_MISPLACED = 2 };
typedef struct HeaacSbrParseInfo {
    int sbr_bits;                 /* num_sbr_bits of ff_decode_sbr_extension, crc and header included */
    int header;                   /* 1: the payload carried a header */
    int ps_present;               /* 1: a PS extension was read */
    int ps_status;                /* HEAAC_PARSE_OK or the PS reader's error (then ps.start = 0) */
} HeaacSbrParseInfo;

/* One SBR payload: `bit` = position in `au` just after the 4-bit extension type (HeaacAacFrameInfo.
 * sbr_payload_bit), cnt = the fill element's byte count, crc = EXT_SBR_DATA_CRC, channels = 1 (SCE) or
 * 2 (CPE), allow_ps = HEAAC_SBR_ALLOW_PS when m4ac.ps != 0 (`ps` may be NULL without it), plus
 * HEAAC_SBR_MISPLACED for a payload the frame parser flagged so (header read, then SBR off: a start = 0 record and
 * HEAAC_PARSE_ERR_DATA).  sample_rate = the AAC core's (the SBR tables are built for twice that, aacsbr.c:1056).
 * Returns HEAAC_PARSE_OK, or a negative error with the records written as described above. */
int heaac_sbr_parse_payload(HeaacSbrStream *st, HeaacSbrHeaderTable *tab, int sample_rate,
                            const uint8_t *au, int size, int bit, int cnt, int crc,
                            int channels, int allow_ps,
                            HeaacSbrFrame *sbr, HeaacPsFrame *ps, HeaacSbrParseInfo *info);

/* The record of an access unit without an SBR payload (start = 0, header and "old" fields kept). */
void heaac_sbr_no_payload(HeaacSbrStream *st, int channels, HeaacSbrFrame *sbr, HeaacPsFrame *ps);

/* A whole HE-AAC access unit: heaac_aac_parse_frame, then the SBR payload it located.
 * Returns the core parser's error if that fails (no SBR record then), else the SBR parser's result
 * (HEAAC_PARSE_NO_SBR when the unit has none). */
int heaac_heaac_parse_frame(const HeaacAacConfig *cfg, HeaacAacStream *st, HeaacSbrStream *sst,
                            HeaacSbrHeaderTable *tab, const uint8_t *au, int size,
                            float *coeffs, HeaacIcs *ics, HeaacToolsFrame *tools,
                            HeaacSbrFrame *sbr, HeaacPsFrame *ps, HeaacAacFrameInfo *info);

/* The same with the channel stride of `coeffs` / `ics` chosen (heaac_aac_parse_frame_ex): 1 packs mono streams. */
int heaac_heaac_parse_frame_ex(const HeaacAacConfig *cfg, HeaacAacStream *st, HeaacSbrStream *sst,
                               HeaacSbrHeaderTable *tab, const uint8_t *au, int size, int coeff_channels,
                               float *coeffs, HeaacIcs *ics, HeaacToolsFrame *tools,
                               HeaacSbrFrame *sbr, HeaacPsFrame *ps, HeaacAacFrameInfo *info);

/* n independent streams, one access unit each, on host threads (see heaac_aac_parse_batch).
 * sbr [n], ps [n] (NULL unless cfg->ps != 0 and the stream is mono). */
int heaac_heaac_parse_batch(const HeaacAacConfig *cfg, HeaacAacStream *st, HeaacSbrStream *sst,
                            HeaacSbrHeaderTable *tab,
                            const uint8_t *const *au, const int *size, size_t n,
                            float *coeffs, HeaacIcs *ics, HeaacToolsFrame *tools,
                            HeaacSbrFrame *sbr, HeaacPsFrame *ps,
                            HeaacAacFrameInfo *info, int *status, int threads);

/* FNV-1a of the generated SBR / PS Huffman tables (sbr_iso_tables.h). */
uint64_t heaac_sbr_tables_fingerprint(void);

#ifdef __cplusplus
}
#endif
#endif /* HEAAC_PARSE_H */
