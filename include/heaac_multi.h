/* heaac_multi.h -- one node, several GPUs: a frame batch sharded by index over the devices of one process
 * (BASELINE config 5: "2 M-frame batch sharded across 8 x MI355X, per-frame independent, xGMI
 * gather of PCM").  Frames are independent once their state is an explicit record (heaac_dsp.h), so device g of G
 * takes the contiguous range heaac_multi_shard() names and nothing is exchanged during compute.  One host thread
 * and one stream per device drive the shards; the only exchange step is the optional gather of the PCM shards
 * into one device's buffer (hipMemcpyPeerAsync over xGMI).
 *
 * The reference has no counterpart (one stream per decoder context, libavcodec/aacdec.c); this is the batched
 * path's own scaling surface.  The Python harness (ffmpeg-heaac_amd/shard.py, bench.py --gpus N) does the same
 * with one PROCESS per device over torch.distributed; this entry is for C hosts.
 */
#ifndef HEAAC_MULTI_H
#define HEAAC_MULTI_H

#include <stddef.h>
#include "heaac_dsp.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct HeaacMulti HeaacMulti;

/* Contiguous, balanced split of n frames over G devices: the first n % G devices take one frame more.
 * Pure arithmetic (no device needed); identical to ffmpeg-heaac_amd/shard.py shard_range(). */
void heaac_multi_shard(size_t n, int g, int G, size_t *first, size_t *count);

/* One HeaacDevice, one stream and one worker thread per entry of devices[] (HIP device ordinals; an ordinal
 * may repeat, which gives two independent contexts on that GPU).  max_frames_per_device sizes each workspace.
 * HEAAC_ERR_NODEVICE without a usable gfx950 device, HEAAC_ERR_ARG for n_devices < 1 or > HEAAC_MULTI_MAX. */
#define HEAAC_MULTI_MAX 16
int heaac_multi_create(HeaacMulti **out, const int *devices, int n_devices, size_t max_frames_per_device);
void heaac_multi_destroy(HeaacMulti *m);
int heaac_multi_devices(const HeaacMulti *m);
/* the context / stream of device slot g (for uploads ordered with the decode) */
HeaacDevice *heaac_multi_device(HeaacMulti *m, int g);
void *heaac_multi_stream(HeaacMulti *m, int g);

/* The arguments of heaac_he_decode_batch_ex for ONE shard, every pointer resident on that shard's device. */
typedef struct HeaacHeShard {
    const float *d_coeffs;
    const HeaacIcs *d_ics;
    const HeaacSbrFrame *d_sbr;
    const HeaacSbrHeader *d_hdr;
    size_t n_hdr;
    const HeaacPsFrame *d_ps;
    const float *d_state_in;
    float *d_state_out;
    void *d_pcm;
    size_t n;                     /* frames of this shard */
} HeaacHeShard;

/* Decode all shards concurrently, shard g on device slot g, and wait for all of them.
 *   gather_pcm    NULL, or a buffer on device slot `gather_slot` that receives the shards' PCM in frame order
 *                 (shard g at the byte offset of frame heaac_multi_shard(...).first), copied device to device
 *                 behind each shard's decode
 * Returns HEAAC_OK or the first failing shard's error. */
int heaac_multi_he_decode(HeaacMulti *m, int cfg, int flags, const HeaacHeShard *shards, int pcm_format,
                          void *gather_pcm, int gather_slot);

#ifdef __cplusplus
}
#endif
#endif /* HEAAC_MULTI_H */
