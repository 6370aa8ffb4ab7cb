/* heaac_pipeline.h -- bitstreams in host memory in, int16 PCM in host memory out: the caller that keeps nothing
 * on the device (SURVEY.md s8f N2 "host-side batched parser ... thousands of concurrent streams").
 *
 * n streams advance in lock step, one access unit per stream and tick.  A tick runs through four stages --
 * host parse (a persistent thread pool), H2D of the parsed records (pinned), spectral tools + decode on the GPU,
 * D2H of the PCM (pinned) -- and consecutive ticks overlap: while tick t is on the link and on the GPU, the host
 * parses tick t + 1.  HEAAC_PIPELINE_DEPTH sets of pinned / device buffers rotate; three HIP streams (copy in,
 * compute, copy out) are ordered by events.  The per-stream decoder state (parser state on the host, DSP state records on the
 * device) lives in the pipeline.
 *
 * What it does per tick is exactly heaac_heaac_parse_frame + heaac_spectral_tools_batch + heaac_he_decode_batch
 * (for plain AAC-LC streams: heaac_aac_parse_frame + tools + heaac_lc_decode_batch), HEAAC_PCM_S16_INTERLEAVED, for
 * every stream; tests compare it with those calls made one after the other.
 * The reference has no counterpart: its decoder handles one packet of one stream per call
 * (avcodec_decode_audio3, libavcodec/utils.c:638-663).
 */
#ifndef HEAAC_PIPELINE_H
#define HEAAC_PIPELINE_H

#include <stddef.h>
#include <stdint.h>
#include "heaac_dsp.h"
#include "heaac_parse.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct HeaacPipeline HeaacPipeline;
#define HEAAC_PIPELINE_DEPTH 4        /* ticks that may be in flight (submitted, not yet collected) */

/* aac: the configuration all streams share (AudioSpecificConfig as heaac_asc_parse leaves it; sbr = 1 for the
 * HE configurations).  AAC-Main streams get their predictor state; an extension rate equal to the core rate is
 * "downsampled SBR" -- 1024 samples per channel and tick at the core rate (aacsbr.c:1719, aacdec.c:2080-2084) --
 * and one strictly between the core rate and twice that is HEAAC_ERR_ARG.
 * he_cfg: HEAAC_CFG_HEV2 (mono core + SBR + PS), HEAAC_CFG_HEV1_MONO, HEAAC_CFG_HEV1 (pair), or HEAAC_CFG_LC_MONO /
 *         HEAAC_CFG_LC_STEREO (no SBR: 1024 samples per channel and tick).
 * threads: parser threads (<= 0: the CPUs the process may use -- online CPUs, capped at twice a cgroup CPU quota;
 *          at most 256).
 * HEAAC_ERR_NODEVICE without a usable device. */
int heaac_pipeline_create(HeaacPipeline **out, const HeaacAacConfig *aac, int he_cfg, size_t n_streams, int threads);
void heaac_pipeline_destroy(HeaacPipeline *p);

/* Tick t: au[n_streams] / size[n_streams], one access unit per stream (the buffers are read during the call
 * only).  Parses on the pool, then enqueues upload, decode and download and returns without waiting for them.
 * status (may be NULL) receives each stream's parse result (heaac_heaac_parse_frame's).  A stream whose access unit
 * does not parse (its core element: a negative status with no channels read; an SBR payload that fails only turns
 * SBR off for the unit, as in the reference) gets SILENCE for this tick and keeps its DSP state record, whether or not
 * status is given (the reference returns an error and writes no samples, aacdec.c:2065-2070).  Window history, noise
 * generator and predictors are left as the reference's decoder leaves its own where the refusal is the reference's
 * (HEAAC_REFUSED_AS_REFERENCE, heaac_parse.h), and as they were before the unit otherwise.  At most
 * HEAAC_PIPELINE_DEPTH ticks may be in flight: one more submit before a collect returns HEAAC_ERR_ARG. */
int heaac_pipeline_submit(HeaacPipeline *p, const uint8_t *const *au, const int *size, int *status);

/* Waits for the OLDEST tick in flight and hands out its PCM: [n_streams][2048 (LC, downsampled SBR: 1024)][channels] int16 in pinned memory
 * owned by the pipeline, valid until HEAAC_PIPELINE_DEPTH more ticks have been submitted. */
int heaac_pipeline_collect(HeaacPipeline *p, const int16_t **pcm);

/* Per-stage wall time of the last collected tick in milliseconds: host parse, H2D, GPU, D2H (device stages
 * from HIP events). */
void heaac_pipeline_timing(const HeaacPipeline *p, float ms[4]);

/* ---- streams of a multi-element layout (aac_decode_frame's element loop, aacdec.c:1999-2076) ----
 * n streams of ONE layout (heaac_aac_layout_default for the channel configurations 1 ... 7, heaac_asc_layout /
 * heaac_aac_layout_from_pce for a program config element) advance in lock step: per tick every stream's access unit is
 * parsed on the pool (heaac_aac_parse_frame_layout_ex + the SBR payload behind each element), the records are laid out
 * element-major, and every element of the layout is ONE batched spectral-tools call and ONE batched decode call over
 * the n streams; heaac_pcm_interleave_batch writes what float_to_int16_interleave writes (:2096-2097).  Exactly what
 * heaac_codec_decode does for one such stream (csrc/codec_layout.hip), which is what the tests compare it with.
 *   aac->sbr: 0 (AAC-LC / Main: 1024 samples per channel and tick) or 1 (explicit SBR per element; 2048, or 1024 for
 *             downsampled SBR).  Implicit signalling (-1) is settled per stream by its first access unit: HEAAC_ERR_ARG.
 *   aac->ps:  with aac->sbr == 1, anything but 0 gives every SINGLE CHANNEL element of the layout a second, Parametric
 *             Stereo output channel directly behind its first, as the reference does for a program-config stream with
 *             explicitly signalled SBR (ps = -1 becomes 1, aacdec.c:476-477; che_configure :203-206; ff_sbr_apply
 *             copies the left channel until PS data arrives, aacsbr.c:1751-1758).  heaac_asc_layout leaves ps = 0 for
 *             the channel configurations 3 ... 7 (mpeg4audio.c:137-139).  heaac_layout_pipeline_channels() says how
 *             many channels a tick's PCM has.
 *   layout:   may name coupling channel elements: each is one more batched individual channel stream per slot -- its
 *             own tools at its place in the stream, dependent coupling around every target's TNS, its own IMDCT (and,
 *             in an SBR stream, its own SBR with the payload behind it) and independent coupling behind the targets'
 *             where it couples AFTER_IMDCT.  The coupling POINT may differ from stream to stream; the coupling
 *             elements' PLACES in the unit must be the same for all streams of a tick.
 * The streams must emit their elements in one bitstream order (the noise generator runs through them in that order;
 * the first good access unit sets it).  A stream whose unit does not parse, leaves an element out or deviates from
 * the order (status HEAAC_PARSE_ERR_UNSUPPORTED) gets silence for the tick and keeps its DSP state; window histories,
 * noise generator and predictors are left as the reference's decoder leaves its own where the refusal is the
 * reference's (as for heaac_pipeline_submit above).
 * Two ticks may be in flight.  PCM: [n_streams][len][heaac_layout_pipeline_channels()] int16, pinned, valid until two
 * more submits. */
typedef struct HeaacLayoutPipeline HeaacLayoutPipeline;
int  heaac_layout_pipeline_create(HeaacLayoutPipeline **out, const HeaacAacConfig *aac, const HeaacAacLayout *layout,
                                  size_t n_streams, int threads);
void heaac_layout_pipeline_destroy(HeaacLayoutPipeline *p);
int  heaac_layout_pipeline_submit(HeaacLayoutPipeline *p, const uint8_t *const *au, const int *size, int *status);
int  heaac_layout_pipeline_collect(HeaacLayoutPipeline *p, const int16_t **pcm);
int  heaac_layout_pipeline_channels(const HeaacLayoutPipeline *p);   /* layout->channels + one per SCE with Parametric Stereo */

#ifdef __cplusplus
}
#endif
#endif /* HEAAC_PIPELINE_H */
