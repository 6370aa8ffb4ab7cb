// capi.hip -- the C ABI of include/heaac_dsp.h: device context + entry points.
//
// Thin by design: argument checks, table upload, kernel launches.  No host
// fallback exists -- if no HIP device is usable every entry point fails with
// HEAAC_ERR_NODEVICE / HEAAC_ERR_HIP.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include "heaac_dsp.h"
#include "heaac_debug.h"
#include "tables.h"
#include "kernels.h"

#define HE_MAX_LANES 4
#define HE_ZERO_BYTES (2 * 38 * 64 * 4)      // one channel's X record of +0: what k_synth reads for bands that were not stored
struct HeaacDevice {
    int device;
    float *d_tab;
    uint16_t *d_rev;
    void *d_work;
    unsigned *d_queue;      // frame-queue heads of the kernels that draw frames dynamically (one set per lane)
    // X hand-over side data: a page of zeros, then per workspace set one byte per frame and channel = the number of QMF
    // bands of the X rows the HF / PS stage has stored (the bands above are +0 and are not written; the synthesis
    // kernel reads them from the zero page instead)
    unsigned char *d_aux;
    size_t work_bytes;
    size_t max_frames;
    size_t chunk;
    int sets;               // workspace sets allocated: 1 (batches of one chunk) or HE_LANES
    // Chunks of one HE call alternate between two internal streams ("lanes"), each with its own
    // workspace half and queue heads: the persistent kernels of one chunk drain while the next
    // chunk's kernels fill the freed CUs, so a chunk can be small enough for its W / X workspace to
    // live in the 256 MiB Infinity Cache between the kernel that writes it and the one that reads it.
    hipStream_t lane[HE_MAX_LANES];
    hipEvent_t fork, join[HE_MAX_LANES];
    int lanes;
    // The HE calls share the workspace and the queue heads, so they must not overlap: the stream of the
    // last HE call and an event behind its launches are kept, and a call that arrives on ANOTHER stream
    // while that work is still in flight is refused (HEAAC_ERR_ARG) instead of racing on the workspace.
    hipStream_t owner;
    hipEvent_t done;
    int owner_valid, done_recorded;
};

// Claim the HE workspace for stream s (see HeaacDevice::owner).
static int he_claim(HeaacDevice *d, hipStream_t s, bool *capturing)
{
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
    *capturing = cs != hipStreamCaptureStatusNone;
    if (d->owner_valid && d->owner != s) {
        // the earlier call's work must have finished; inside a capture that cannot be asked
        if (*capturing) return HEAAC_ERR_ARG;
        if (d->done_recorded) {
            const hipError_t q = hipEventQuery(d->done);
            if (q == hipErrorNotReady) return HEAAC_ERR_ARG;
            if (q != hipSuccess) { (void)hipGetLastError(); return HEAAC_ERR_HIP; }
        }
    }
    d->owner = s;
    d->owner_valid = 1;
    return HEAAC_OK;
}
static int he_release(HeaacDevice *d, hipStream_t s, bool capturing, int rc)
{
    d->done_recorded = 0;
    if (!capturing) {
        if (hipEventRecord(d->done, s) != hipSuccess) return rc == HEAAC_OK ? HEAAC_ERR_HIP : rc;
        d->done_recorded = 1;
    }
    return rc;
}
static int he_lanes()
{
    int l = 2;
#ifdef HEAAC_TUNING
    const char *env = getenv("HEAAC_LANES");                 /* -DHEAAC_TUNING builds only: lanes sweep */
    if (env) l = atoi(env);
#endif
    return l < 1 ? 1 : l > HE_MAX_LANES ? HE_MAX_LANES : l;
}
#define HE_LANES he_lanes()

extern "C" const char *heaac_build_info(void)
{
    return "heaac-amd gfx950 (hipcc, -ffp-contract=off) " __DATE__ " " __TIME__;
}

extern "C" const char *heaac_strerror(int err)
{
    switch (err) {
    case HEAAC_OK: return "ok";
    case HEAAC_ERR_ARG: return "bad argument or unsupported configuration";
    case HEAAC_ERR_HIP: return "HIP runtime error";
    case HEAAC_ERR_NOMEM: return "out of memory";
    case HEAAC_ERR_NODEVICE: return "no usable gfx950 device";
    }
    return "unknown error";
}

// HE pipeline stages exchange W[ncore][32][32][2] and X[2][2][38][64] per frame through this
// workspace, one chunk of frames at a time.  Measured on MI355X (profiles/r01_chunk_sweep.txt,
// profiles/r02_chunk_lanes.txt): throughput rises with the chunk size (the tails of the persistent
// kernels amortise) up to the whole 256 k-frame batch; chunks small enough to keep the workspace
// in the 256 MiB Infinity Cache lose more in tails than they save, also when consecutive chunks
// overlap on two streams.  Batches beyond one chunk alternate between two lanes (below).
#define HE_CHUNK_FRAMES 262144
#define WS_W_FLOATS (2 * 2048)
#define WS_X_FLOATS (2 * 2 * 38 * 64)

static size_t he_chunk_frames(size_t max_frames)
{
    size_t cap = HE_CHUNK_FRAMES;
#ifdef HEAAC_TUNING
    const char *env = getenv("HEAAC_CHUNK_FRAMES");          /* -DHEAAC_TUNING builds only: chunk sweep */
    if (env && atol(env) >= 64) cap = (size_t)atol(env);
#endif
    size_t chunk = max_frames < cap ? max_frames : cap;
    return chunk < 64 ? 64 : chunk;
}

extern "C" size_t heaac_device_workspace_bytes(size_t max_frames)
{
    // one workspace set per lane; a batch that fits one chunk runs on the caller's stream alone
    const size_t chunk = he_chunk_frames(max_frames);
    size_t sets = max_frames > chunk ? (size_t)HE_LANES : 1;
    if (sets > 1 && (max_frames + chunk - 1) / chunk < sets) sets = (max_frames + chunk - 1) / chunk;
    return sets * chunk * (WS_W_FLOATS + WS_X_FLOATS) * sizeof(float);
}

extern "C" int heaac_device_create(HeaacDevice **out, size_t max_frames)
{
    if (!out) return HEAAC_ERR_ARG;
    *out = NULL;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return HEAAC_ERR_NODEVICE;
    HeaacDevice *d = (HeaacDevice *)calloc(1, sizeof(*d));
    if (!d) return HEAAC_ERR_NOMEM;
    if (hipGetDevice(&d->device) != hipSuccess) { free(d); return HEAAC_ERR_HIP; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, d->device) != hipSuccess ||
        strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        free(d);
        return HEAAC_ERR_NODEVICE;
    }
    HeaacHostTables *t = (HeaacHostTables *)malloc(sizeof(*t));
    if (!t) { free(d); return HEAAC_ERR_NOMEM; }
    heaac_build_tables(t);
    d->max_frames = max_frames;
    d->work_bytes = heaac_device_workspace_bytes(max_frames);
    d->chunk = he_chunk_frames(max_frames);
    d->sets = max_frames > d->chunk ? HE_LANES : 1;
    if (d->sets > 1 && (max_frames + d->chunk - 1) / d->chunk < (size_t)d->sets)
        d->sets = (int)((max_frames + d->chunk - 1) / d->chunk);
    int rc = HEAAC_OK;
    if (hipMalloc((void **)&d->d_tab, sizeof(t->f)) != hipSuccess ||
        hipMalloc((void **)&d->d_rev, sizeof(t->rev)) != hipSuccess ||
        hipMalloc((void **)&d->d_queue, 64 * HE_MAX_LANES) != hipSuccess ||
        hipMalloc((void **)&d->d_aux, HE_ZERO_BYTES + 2 * (size_t)(d->sets > 0 ? d->sets : 1) * (d->chunk ? d->chunk : 1)) != hipSuccess ||
        (d->work_bytes && hipMalloc(&d->d_work, d->work_bytes) != hipSuccess))
        rc = HEAAC_ERR_NOMEM;
    if (rc == HEAAC_OK &&
        (hipMemset(d->d_aux, 0, HE_ZERO_BYTES) != hipSuccess ||
         hipMemcpy(d->d_tab, t->f, sizeof(t->f), hipMemcpyHostToDevice) != hipSuccess ||
         hipMemcpy(d->d_rev, t->rev, sizeof(t->rev), hipMemcpyHostToDevice) != hipSuccess))
        rc = HEAAC_ERR_HIP;
    free(t);
    d->lanes = HE_LANES;
    if (rc == HEAAC_OK && (hipEventCreateWithFlags(&d->fork, hipEventDisableTiming) != hipSuccess ||
                           hipEventCreateWithFlags(&d->done, hipEventDisableTiming) != hipSuccess))
        rc = HEAAC_ERR_HIP;
    for (int k = 0; k < d->lanes && rc == HEAAC_OK; k++)
        if (hipEventCreateWithFlags(&d->join[k], hipEventDisableTiming) != hipSuccess ||
            hipStreamCreateWithFlags(&d->lane[k], hipStreamNonBlocking) != hipSuccess)
            rc = HEAAC_ERR_HIP;
    if (rc != HEAAC_OK) {
        heaac_device_destroy(d);
        return rc;
    }
    *out = d;
    return HEAAC_OK;
}

extern "C" void heaac_device_destroy(HeaacDevice *d)
{
    if (!d) return;
    if (d->d_tab) (void)hipFree(d->d_tab);
    if (d->d_rev) (void)hipFree(d->d_rev);
    if (d->d_work) (void)hipFree(d->d_work);
    if (d->d_queue) (void)hipFree(d->d_queue);
    if (d->d_aux) (void)hipFree(d->d_aux);
    for (int k = 0; k < HE_MAX_LANES; k++) {
        if (d->lane[k]) (void)hipStreamDestroy(d->lane[k]);
        if (d->join[k]) (void)hipEventDestroy(d->join[k]);
    }
    if (d->fork) (void)hipEventDestroy(d->fork);
    if (d->done) (void)hipEventDestroy(d->done);
    free(d);
}

extern "C" int heaac_imdct_half_batch(HeaacDevice *dev, int which, float *d_out, const float *d_in,
                                      size_t n, void *stream)
{
    if (!dev || which < 0 || which > 3)
        return HEAAC_ERR_ARG;
    if (n == 0)
        return HEAAC_OK;
    if (!d_out || !d_in || d_out == d_in)
        return HEAAC_ERR_ARG;
    return heaac_launch_imdct_half(dev->d_tab, dev->d_rev, which, d_out, d_in, n, (hipStream_t)stream);
}

extern "C" int heaac_lc_decode_batch(HeaacDevice *dev, int channels,
                                     const float *d_coeffs, const HeaacIcs *d_ics,
                                     const float *d_state_in, float *d_state_out,
                                     void *d_pcm, int pcm_format, size_t n, void *stream)
{
    if (!dev || channels < 1 || channels > 2 ||
        (pcm_format != HEAAC_PCM_F32_PLANAR && pcm_format != HEAAC_PCM_S16_INTERLEAVED &&
         pcm_format != HEAAC_PCM_S16_INTERLEAVED_SSE2))
        return HEAAC_ERR_ARG;
    if (n == 0)
        return HEAAC_OK;
    if (!d_coeffs || !d_ics || !d_state_in || !d_state_out || !d_pcm)
        return HEAAC_ERR_ARG;
    return heaac_launch_lc(dev->d_tab, dev->d_rev, channels, d_coeffs, d_ics, d_state_in, d_state_out,
                           d_pcm, pcm_format, n, (hipStream_t)stream);
}

extern "C" int heaac_spectral_tools_batch(HeaacDevice *dev, int channels, float *d_coeffs,
                                          const HeaacToolsFrame *d_tools,
                                          const int32_t *d_rng_in, int32_t *d_rng_out,
                                          const HeaacPredictorState *d_pred_in, HeaacPredictorState *d_pred_out,
                                          size_t n, void *stream)
{
    return heaac_spectral_tools_batch_ex(dev, channels, HEAAC_TOOLS_ALL, d_coeffs, d_tools, d_rng_in, d_rng_out,
                                         d_pred_in, d_pred_out, NULL, NULL, 0, n, stream);
}

extern "C" int heaac_spectral_tools_batch_ex(HeaacDevice *dev, int channels, int stages, float *d_coeffs,
                                             const HeaacToolsFrame *d_tools,
                                             const int32_t *d_rng_in, int32_t *d_rng_out,
                                             const HeaacPredictorState *d_pred_in, HeaacPredictorState *d_pred_out,
                                             const HeaacCceFrame *d_cce, const float *d_cce_coeffs, int n_cce,
                                             size_t n, void *stream)
{
    if (!dev || channels < 1 || channels > 2 || (d_rng_in && !d_rng_out) || (d_pred_in && !d_pred_out) ||
        !stages || (stages & ~HEAAC_TOOLS_ALL) || n_cce < 0 || n_cce > HEAAC_MAX_CCE ||
        (n_cce && (!d_cce || !d_cce_coeffs)))
        return HEAAC_ERR_ARG;
    if (n == 0)
        return HEAAC_OK;
    if (!d_coeffs || !d_tools)
        return HEAAC_ERR_ARG;
    return heaac_launch_spectral_tools(channels, d_coeffs, d_tools, d_rng_in, d_rng_out, d_pred_in, d_pred_out,
                                       stages, d_cce, d_cce_coeffs, n_cce, n, (hipStream_t)stream);
}

extern "C" int heaac_he_decode_batch_ex(HeaacDevice *dev, int cfg, int flags,
                                        const float *d_coeffs, const HeaacIcs *d_ics,
                                        const HeaacSbrFrame *d_sbr,
                                        const HeaacSbrHeader *d_hdr, size_t n_hdr,
                                        const HeaacPsFrame *d_ps,
                                        const float *d_state_in, float *d_state_out,
                                        void *d_pcm, int pcm_format,
                                        size_t n, void *stream);

extern "C" int heaac_he_decode_batch(HeaacDevice *dev, int cfg,
                                     const float *d_coeffs, const HeaacIcs *d_ics,
                                     const HeaacSbrFrame *d_sbr,
                                     const HeaacSbrHeader *d_hdr, size_t n_hdr,
                                     const HeaacPsFrame *d_ps,
                                     const float *d_state_in, float *d_state_out,
                                     void *d_pcm, int pcm_format,
                                     size_t n, void *stream)
{
    return heaac_he_decode_batch_ex(dev, cfg, 0, d_coeffs, d_ics, d_sbr, d_hdr, n_hdr, d_ps, d_state_in, d_state_out,
                                    d_pcm, pcm_format, n, stream);
}

extern "C" int heaac_he_decode_batch_ex(HeaacDevice *dev, int cfg, int flags,
                                     const float *d_coeffs, const HeaacIcs *d_ics,
                                     const HeaacSbrFrame *d_sbr,
                                     const HeaacSbrHeader *d_hdr, size_t n_hdr,
                                     const HeaacPsFrame *d_ps,
                                     const float *d_state_in, float *d_state_out,
                                     void *d_pcm, int pcm_format,
                                     size_t n, void *stream)
{
    if (!dev || (cfg != HEAAC_CFG_HEV1 && cfg != HEAAC_CFG_HEV1_MONO && cfg != HEAAC_CFG_HEV2) ||
        (pcm_format != HEAAC_PCM_F32_PLANAR && pcm_format != HEAAC_PCM_S16_INTERLEAVED &&
         pcm_format != HEAAC_PCM_S16_INTERLEAVED_SSE2) ||
        (flags & ~HEAAC_HE_DOWNSAMPLED))
        return HEAAC_ERR_ARG;
    if (n == 0)
        return HEAAC_OK;
    if (!d_coeffs || !d_ics || !d_sbr || !d_hdr || !n_hdr || !d_state_in || !d_state_out || !d_pcm ||
        (cfg == HEAAC_CFG_HEV2 && !d_ps))
        return HEAAC_ERR_ARG;
    const int ncore = cfg == HEAAC_CFG_HEV1 ? 2 : 1;
    const int nout  = cfg == HEAAC_CFG_HEV1_MONO ? 1 : 2;
    const size_t words = cfg == HEAAC_CFG_HEV1 ? HEAAC_STATE_WORDS_HEV1 :
                         cfg == HEAAC_CFG_HEV2 ? HEAAC_STATE_WORDS_HEV2 : HEAAC_STATE_WORDS_HEV1_MONO;
    const size_t pcm_bytes = (size_t)nout * ((flags & HEAAC_HE_DOWNSAMPLED) ? 1024 : 2048) *
                             (pcm_format == HEAAC_PCM_F32_PLANAR ? 4 : 2);
    const size_t set_floats = dev->chunk * (WS_W_FLOATS + WS_X_FLOATS);
    hipStream_t s = (hipStream_t)stream;
    bool capturing = false;
    const int claim = he_claim(dev, s, &capturing);
    if (claim != HEAAC_OK) return claim;
    // one chunk: everything on the caller's stream; more: fork onto the two lanes and join again
    // (event fork / join, so the call stays capturable into a hipGraph)
    const bool lanes = n > dev->chunk && dev->sets > 1;
    const int nl = lanes ? dev->sets : 1;
    // (every exit behind he_claim goes through the join and he_release below: a failed fork must not leave the
    // workspace claimed or lanes that were already forked unjoined)
    int rc = HEAAC_OK;
    int forked = 0;
    if (lanes) {
        if (hipEventRecord(dev->fork, s) != hipSuccess) rc = HEAAC_ERR_HIP;
        for (int k = 0; k < nl && rc == HEAAC_OK; k++) {
            if (hipStreamWaitEvent(dev->lane[k], dev->fork, 0) != hipSuccess) rc = HEAAC_ERR_HIP;
            else forked = k + 1;
        }
    }
    size_t c = 0;
    for (size_t f0 = 0; f0 < n && rc == HEAAC_OK; f0 += dev->chunk, c++) {
        const size_t nc = n - f0 < dev->chunk ? n - f0 : dev->chunk;
        const int k = (int)(c % nl);
        float *ws_W = (float *)dev->d_work + k * set_floats;
        float *ws_X = ws_W + dev->chunk * WS_W_FLOATS;
        rc = heaac_launch_he(dev->d_tab, dev->d_rev, cfg,
                             d_coeffs + f0 * ncore * 1024, d_ics + f0 * ncore,
                             d_sbr + f0, d_hdr, (unsigned)(n_hdr > 0xffffu ? 0x10000u : n_hdr), d_ps ? d_ps + f0 : NULL,
                             d_state_in + f0 * words, d_state_out + f0 * words,
                             (char *)d_pcm + f0 * pcm_bytes, pcm_format,
                             ws_W, ws_X, dev->d_queue + 16 * k,
                             dev->d_aux + HE_ZERO_BYTES + 2 * (size_t)k * dev->chunk, (const float *)dev->d_aux,
                             nc, 0, flags, lanes ? dev->lane[k] : s);
    }
    if (lanes) {
        // always rejoin, also after a failed launch (a capture must not be left forked)
        for (int k = 0; k < forked; k++)
            if (hipEventRecord(dev->join[k], dev->lane[k]) != hipSuccess ||
                hipStreamWaitEvent(s, dev->join[k], 0) != hipSuccess)
                rc = rc == HEAAC_OK ? HEAAC_ERR_HIP : rc;
    }
    return he_release(dev, s, capturing, rc);
}

extern "C" int heaac_qmf_analysis_batch(HeaacDevice *dev, const float *d_in,
                                        const float *d_xhist_in, float *d_xhist_out,
                                        float *d_W, float scale, size_t n, void *stream)
{
    if (!dev) return HEAAC_ERR_ARG;
    if (n == 0) return HEAAC_OK;
    if (!d_in || !d_xhist_in || !d_xhist_out || !d_W) return HEAAC_ERR_ARG;
    return heaac_launch_qmf_analysis(dev->d_tab, d_in, d_xhist_in, d_xhist_out, d_W, scale, n,
                                     (hipStream_t)stream);
}

extern "C" int heaac_qmf_synthesis_batch(HeaacDevice *dev, const float *d_X,
                                         const float *d_v_in, float *d_v_out,
                                         float *d_out, float scale, float bias,
                                         size_t n, void *stream)
{
    if (!dev) return HEAAC_ERR_ARG;
    if (n == 0) return HEAAC_OK;
    if (!d_X || !d_v_in || !d_v_out || !d_out) return HEAAC_ERR_ARG;
    return heaac_launch_qmf_synthesis(dev->d_tab, d_X, d_v_in, d_v_out, d_out, scale, bias, n,
                                      (hipStream_t)stream);
}

extern "C" int heaac_qmf_synthesis_ds_batch(HeaacDevice *dev, const float *d_X,
                                            const float *d_v_in, float *d_v_out,
                                            float *d_out, float scale, float bias,
                                            size_t n, void *stream)
{
    if (!dev) return HEAAC_ERR_ARG;
    if (n == 0) return HEAAC_OK;
    if (!d_X || !d_v_in || !d_v_out || !d_out) return HEAAC_ERR_ARG;
    return heaac_launch_qmf_synthesis_ds(dev->d_tab, d_X, d_v_in, d_v_out, d_out, scale, bias, n,
                                         (hipStream_t)stream);
}

extern "C" int heaac_couple_after_imdct_batch(HeaacDevice *dev, int channels, float *d_pcm, const float *d_cce,
                                              const HeaacCoupling *d_coupling, int16_t *d_s16,
                                              size_t n, void *stream)
{
    if (!dev || (channels != 1 && channels != 2)) return HEAAC_ERR_ARG;
    if (n == 0) return HEAAC_OK;
    if (!d_pcm || !d_cce || !d_coupling || n > 0x7fffffffu) return HEAAC_ERR_ARG;
    return heaac_launch_couple(channels, d_pcm, d_cce, d_coupling, d_s16, n, (hipStream_t)stream);
}

extern "C" int heaac_pcm_interleave_batch(HeaacDevice *dev, int channels, const HeaacPlaneRef *planes, int len,
                                          int pcm_format, int16_t *d_out, size_t n, void *stream)
{
    if (!dev || channels < 1 || channels > HEAAC_MAX_PCM_PLANES || len <= 0 || (len & 3) ||
        (pcm_format != HEAAC_PCM_S16_INTERLEAVED && pcm_format != HEAAC_PCM_S16_INTERLEAVED_SSE2))
        return HEAAC_ERR_ARG;
    if (n == 0) return HEAAC_OK;
    if (!planes || !d_out) return HEAAC_ERR_ARG;
    for (int c = 0; c < channels; c++)
        if (!planes[c].d_base || ((uintptr_t)planes[c].d_base & 15) || (planes[c].frame_stride & 3)) return HEAAC_ERR_ARG;
    return heaac_launch_interleave(channels, planes, len, pcm_format, d_out, n, (hipStream_t)stream);
}

// include/heaac_debug.h: device pointers of workspace set 0
extern "C" int heaac_debug_workspace(HeaacDevice *dev, float **d_W, float **d_X, size_t *chunk)
{
    if (!dev) return HEAAC_ERR_ARG;
    if (d_W) *d_W = (float *)dev->d_work;
    if (d_X) *d_X = (float *)dev->d_work + dev->chunk * WS_W_FLOATS;      /* set 0 */
    if (chunk) *chunk = dev->chunk;
    return HEAAC_OK;
}

extern "C" int heaac_debug_xbands(HeaacDevice *dev, unsigned char *host_out, size_t n_frames)
{
    if (!dev || !host_out || n_frames > dev->chunk) return HEAAC_ERR_ARG;
    if (hipDeviceSynchronize() != hipSuccess) return HEAAC_ERR_HIP;
    return hipMemcpy(host_out, dev->d_aux + HE_ZERO_BYTES, 2 * n_frames, hipMemcpyDeviceToHost) == hipSuccess
               ? HEAAC_OK : HEAAC_ERR_HIP;
}

// internal: device table pointers for shim.hip
extern "C" const float *heaac_device_tables(HeaacDevice *dev, const uint16_t **rev)
{
    if (rev) *rev = dev->d_rev;
    return dev->d_tab;
}
