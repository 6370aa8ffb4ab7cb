// multi.hip -- include/heaac_multi.h: a frame batch over the GPUs of one node from one process.
//
// One worker thread per device slot owns that device's HIP context for the lifetime of the HeaacMulti
// (hipSetDevice is per thread), its HeaacDevice and its stream.  A decode call hands every worker its shard,
// the workers launch, optionally copy their PCM shard into the gather buffer (peer copy) and synchronise
// their stream; the caller returns when all have reported.
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include "heaac_multi.h"

extern "C" void heaac_multi_shard(size_t n, int g, int G, size_t *first, size_t *count)
{
    if (G < 1 || g < 0 || g >= G) { if (first) *first = 0; if (count) *count = 0; return; }
    const size_t base = n / (size_t)G, extra = n % (size_t)G, ug = (size_t)g;
    if (first) *first = ug * base + (ug < extra ? ug : extra);
    if (count) *count = base + (ug < extra ? 1 : 0);
}

struct Job {
    int cfg, flags, pcm_format;
    HeaacHeShard shard;
    void *gather_dst;              // destination of this shard inside the gather buffer, or NULL
    size_t gather_bytes;
    int gather_device;
};

struct Slot {
    HeaacMulti *owner;
    int index, device;
    pthread_t thread;
    HeaacDevice *dev;
    hipStream_t stream;
    int init_rc;
    // hand-over
    Job job;
    int have_job, quit, rc;
};

struct HeaacMulti {
    int n;
    size_t max_frames;
    pthread_mutex_t mu;
    pthread_cond_t cv_work, cv_done;
    int pending, started;
    Slot slot[HEAAC_MULTI_MAX];
};

static int run_job(Slot *s)
{
    const Job &j = s->job;
    const HeaacHeShard &h = j.shard;
    if (h.n == 0) return HEAAC_OK;
    int rc = heaac_he_decode_batch_ex(s->dev, j.cfg, j.flags, h.d_coeffs, h.d_ics, h.d_sbr, h.d_hdr, h.n_hdr, h.d_ps,
                                      h.d_state_in, h.d_state_out, h.d_pcm, j.pcm_format, h.n, (void *)s->stream);
    if (rc == HEAAC_OK && j.gather_dst && j.gather_dst != h.d_pcm) {
        // behind the decode on the shard's own stream: device to device, over xGMI when the devices differ
        const hipError_t e = j.gather_device == s->device
            ? hipMemcpyAsync(j.gather_dst, h.d_pcm, j.gather_bytes, hipMemcpyDeviceToDevice, s->stream)
            : hipMemcpyPeerAsync(j.gather_dst, j.gather_device, h.d_pcm, s->device, j.gather_bytes, s->stream);
        if (e != hipSuccess) rc = HEAAC_ERR_HIP;
    }
    if (hipStreamSynchronize(s->stream) != hipSuccess && rc == HEAAC_OK) rc = HEAAC_ERR_HIP;
    return rc;
}

static void *worker(void *arg)
{
    Slot *s = (Slot *)arg;
    HeaacMulti *m = s->owner;
    int rc = HEAAC_OK;
    if (hipSetDevice(s->device) != hipSuccess) rc = HEAAC_ERR_NODEVICE;
    if (rc == HEAAC_OK) rc = heaac_device_create(&s->dev, m->max_frames);
    if (rc == HEAAC_OK && hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) rc = HEAAC_ERR_HIP;
    pthread_mutex_lock(&m->mu);
    s->init_rc = rc;
    m->started++;
    pthread_cond_broadcast(&m->cv_done);
    for (;;) {
        while (!s->have_job && !s->quit) pthread_cond_wait(&m->cv_work, &m->mu);
        if (s->quit) break;
        pthread_mutex_unlock(&m->mu);
        const int r = s->init_rc == HEAAC_OK ? run_job(s) : s->init_rc;
        pthread_mutex_lock(&m->mu);
        s->rc = r;
        s->have_job = 0;
        m->pending--;
        pthread_cond_broadcast(&m->cv_done);
    }
    pthread_mutex_unlock(&m->mu);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    if (s->dev) heaac_device_destroy(s->dev);
    return NULL;
}

extern "C" int heaac_multi_create(HeaacMulti **out, const int *devices, int n_devices, size_t max_frames_per_device)
{
    if (!out) return HEAAC_ERR_ARG;
    *out = NULL;
    if (!devices || n_devices < 1 || n_devices > HEAAC_MULTI_MAX) return HEAAC_ERR_ARG;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return HEAAC_ERR_NODEVICE;
    for (int g = 0; g < n_devices; g++)
        if (devices[g] < 0 || devices[g] >= count) return HEAAC_ERR_ARG;
    HeaacMulti *m = (HeaacMulti *)calloc(1, sizeof(*m));
    if (!m) return HEAAC_ERR_NOMEM;
    m->n = n_devices;
    m->max_frames = max_frames_per_device;
    pthread_mutex_init(&m->mu, NULL);
    pthread_cond_init(&m->cv_work, NULL);
    pthread_cond_init(&m->cv_done, NULL);
    int launched = 0;
    for (int g = 0; g < n_devices; g++) {
        Slot *s = &m->slot[g];
        s->owner = m; s->index = g; s->device = devices[g];
        if (pthread_create(&s->thread, NULL, worker, s) != 0) break;
        launched++;
    }
    pthread_mutex_lock(&m->mu);
    while (m->started < launched) pthread_cond_wait(&m->cv_done, &m->mu);
    pthread_mutex_unlock(&m->mu);
    int rc = launched == n_devices ? HEAAC_OK : HEAAC_ERR_NOMEM;
    for (int g = 0; g < launched && rc == HEAAC_OK; g++) rc = m->slot[g].init_rc;
    // every device must reach the gather target and vice versa (no-op when already enabled or the same device)
    if (rc == HEAAC_OK) {
        // (on the caller's thread: its current device is put back afterwards)
        int caller_device = -1;
        if (hipGetDevice(&caller_device) != hipSuccess) { (void)hipGetLastError(); caller_device = -1; }
        for (int a = 0; a < n_devices; a++)
            for (int b = 0; b < n_devices; b++) {
                int can = 0;
                if (devices[a] == devices[b]) continue;
                if (hipDeviceCanAccessPeer(&can, devices[a], devices[b]) == hipSuccess && can) {
                    if (hipSetDevice(devices[a]) == hipSuccess) {
                        const hipError_t e = hipDeviceEnablePeerAccess(devices[b], 0);
                        if (e != hipSuccess) (void)hipGetLastError();      // already enabled
                    }
                }
            }
        if (caller_device >= 0) (void)hipSetDevice(caller_device);
    }
    if (rc != HEAAC_OK) {
        m->n = launched;
        heaac_multi_destroy(m);
        return rc;
    }
    *out = m;
    return HEAAC_OK;
}

extern "C" void heaac_multi_destroy(HeaacMulti *m)
{
    if (!m) return;
    pthread_mutex_lock(&m->mu);
    for (int g = 0; g < m->n; g++) m->slot[g].quit = 1;
    pthread_cond_broadcast(&m->cv_work);
    pthread_mutex_unlock(&m->mu);
    for (int g = 0; g < m->n; g++) pthread_join(m->slot[g].thread, NULL);
    pthread_cond_destroy(&m->cv_work);
    pthread_cond_destroy(&m->cv_done);
    pthread_mutex_destroy(&m->mu);
    free(m);
}

extern "C" int heaac_multi_devices(const HeaacMulti *m) { return m ? m->n : 0; }
extern "C" HeaacDevice *heaac_multi_device(HeaacMulti *m, int g) { return m && g >= 0 && g < m->n ? m->slot[g].dev : NULL; }
extern "C" void *heaac_multi_stream(HeaacMulti *m, int g) { return m && g >= 0 && g < m->n ? (void *)m->slot[g].stream : NULL; }

static size_t pcm_frame_bytes(int cfg, int flags, int pcm_format)
{
    const size_t nout = cfg == HEAAC_CFG_HEV1_MONO ? 1 : 2;
    return nout * ((flags & HEAAC_HE_DOWNSAMPLED) ? 1024 : 2048) * (pcm_format == HEAAC_PCM_F32_PLANAR ? 4 : 2);
}

extern "C" int heaac_multi_he_decode(HeaacMulti *m, int cfg, int flags, const HeaacHeShard *shards, int pcm_format,
                                     void *gather_pcm, int gather_slot)
{
    if (!m || !shards || (gather_pcm && (gather_slot < 0 || gather_slot >= m->n))) return HEAAC_ERR_ARG;
    if (cfg != HEAAC_CFG_HEV1 && cfg != HEAAC_CFG_HEV1_MONO && cfg != HEAAC_CFG_HEV2) return HEAAC_ERR_ARG;
    const size_t fb = pcm_frame_bytes(cfg, flags, pcm_format);
    size_t first = 0;
    pthread_mutex_lock(&m->mu);
    for (int g = 0; g < m->n; g++) {
        Slot *s = &m->slot[g];
        s->job.cfg = cfg; s->job.flags = flags; s->job.pcm_format = pcm_format;
        s->job.shard = shards[g];
        s->job.gather_dst = gather_pcm ? (char *)gather_pcm + first * fb : NULL;
        s->job.gather_bytes = shards[g].n * fb;
        s->job.gather_device = gather_pcm ? m->slot[gather_slot].device : -1;
        first += shards[g].n;
        s->have_job = 1;
        s->rc = HEAAC_OK;
    }
    m->pending = m->n;
    pthread_cond_broadcast(&m->cv_work);
    while (m->pending > 0) pthread_cond_wait(&m->cv_done, &m->mu);
    int rc = HEAAC_OK;
    for (int g = 0; g < m->n && rc == HEAAC_OK; g++) rc = m->slot[g].rc;
    pthread_mutex_unlock(&m->mu);
    return rc;
}
