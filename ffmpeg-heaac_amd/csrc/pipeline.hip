// pipeline.hip -- include/heaac_pipeline.h: access units in host memory -> int16 PCM in host memory, the four
// stages of consecutive ticks overlapped (host parse || H2D || GPU || D2H).
//
//   host parse   persistent pool: worker w owns the streams [n w / W, n (w + 1) / W) (their parser state too)
//   H2D          stream `in`:  parsed records of the tick's buffer set, pinned -> device
//   GPU          stream `run`: heaac_spectral_tools_batch + heaac_he_decode_batch, DSP state in place
//   D2H          stream `out`: int16 PCM of the tick's buffer set, device -> pinned
// PL_DEPTH buffer sets rotate (set = tick % PL_DEPTH): a tick spends parse + H2D + GPU + D2H in flight (about 24 ms for
// 32 k streams) while the slowest stage takes 7 ms, so four ticks must overlap to keep every stage busy.
// Event order per set s:  in waits run_done[s] of the tick that used s last (its inputs are free again); run waits
// in_done[s] and that tick's out_done[s] (its PCM buffer is free); out waits run_done[s].
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <alloca.h>
#include <unistd.h>
#include "heaac_pipeline.h"
#include "codec_layout.h"        // heaac_sbr_output_mode

#define PL_MAX_HDRS 4096
#define PL_DEPTH HEAAC_PIPELINE_DEPTH

struct Set {
    // pinned host
    float *h_coeffs; HeaacIcs *h_ics; HeaacToolsFrame *h_tools; HeaacSbrFrame *h_sbr; HeaacPsFrame *h_ps; int16_t *h_pcm;
    // device
    float *d_coeffs; HeaacIcs *d_ics; HeaacToolsFrame *d_tools; HeaacSbrFrame *d_sbr; HeaacPsFrame *d_ps; int16_t *d_pcm;
    hipEvent_t in_start, in_done, run_done, out_done;
    int used;                       // a tick has gone through this set
    float parse_ms;
    unsigned char *failed;          // [n] the stream's access unit of this tick did not parse (core element): 1, or 2 where
                                    // the spectral tools still have to move its noise generator / predictors
    unsigned *h_list, *d_list;      // [2 n] (stream, parking row) pairs of the failed streams, pinned / device
};

// Rows of the failed streams to the parking area and back (or zeroed): one block per listed stream.  A tick with
// thousands of damaged units costs a handful of launches, not six copies per stream (tools/damage_rate.py).
//   mode 0: park[row] = rows[stream];  1: rows[stream] = park[row];  2: rows[stream] = 0
__global__ void k_rows(const unsigned *__restrict__ list, unsigned *rows, unsigned *park, unsigned long long row_words, int mode)
{
    const unsigned stream = list[2 * blockIdx.x], slot = list[2 * blockIdx.x + 1];
    unsigned *r = rows + stream * row_words;
    unsigned *q = park ? park + slot * row_words : nullptr;
    for (unsigned long long w = threadIdx.x; w < row_words; w += blockDim.x) {
        if (mode == 0) q[w] = r[w];
        else if (mode == 1) r[w] = q[w];
        else r[w] = 0u;
    }
}

struct HeaacPipeline {
    HeaacAacConfig aac;
    int he_cfg, ncore, nout, he, out_len;
    size_t n, words;
    HeaacDevice *dev;
    hipStream_t in, run, out;
    Set set[PL_DEPTH];
    float *d_state; int32_t *d_rng;
    // A stream whose access unit fails keeps its decoder state and gets silence for the tick: its state
    // rows are parked here around the decode launches.  Grown on demand -- damaged units are the exception.
    float *d_park_state; int32_t *d_park_rng; HeaacPredictorState *d_park_pred; size_t park_rows;
    HeaacPredictorState *d_pred;  // AAC-Main streams: [n][ncore][672] (aacdec.c:1271-1322), else NULL
    int downsampled;              // SBR with the output at the core rate (aacsbr.c:1719)
    HeaacSbrHeader *d_hdr; size_t hdr_uploaded;
    HeaacSbrHeaderTable *tab;
    HeaacAacStream *ast; HeaacSbrStream *sst;
    unsigned long submitted, collected;
    float last_ms[4];
    // pool
    int threads;
    pthread_t *tid;
    pthread_mutex_t mu;
    pthread_cond_t cv_go, cv_done;
    unsigned long generation;
    int pending, quit;
    const uint8_t *const *job_au; const int *job_size; int *job_status; Set *job_set;
};

struct WorkerArg { HeaacPipeline *p; int w; };

static void parse_slice(HeaacPipeline *p, int w)
{
    const size_t lo = p->n * (size_t)w / (size_t)p->threads, hi = p->n * (size_t)(w + 1) / (size_t)p->threads;
    Set *s = p->job_set;
    for (size_t i = lo; i < hi; i++) {
        HeaacAacFrameInfo fi;
        memset(&fi, 0, sizeof(fi));
        HeaacSbrStream *sst_i = (HeaacSbrStream *)((char *)p->sst + i * heaac_sbr_stream_bytes());
        const int r = p->he
            ? heaac_heaac_parse_frame_ex(&p->aac, &p->ast[i], sst_i, p->tab, p->job_au[i], p->job_size[i],
                                         p->ncore, s->h_coeffs + i * (size_t)p->ncore * 1024, s->h_ics + i * p->ncore,
                                         &s->h_tools[i], &s->h_sbr[i], s->h_ps ? &s->h_ps[i] : NULL, &fi)
            : heaac_aac_parse_frame_ex(&p->aac, &p->ast[i], p->job_au[i], p->job_size[i], p->ncore,
                                       s->h_coeffs + i * (size_t)p->ncore * 1024, s->h_ics + i * p->ncore, &s->h_tools[i],
                                       NULL, &fi);
        if (p->job_status) p->job_status[i] = r;
        // The core element did not parse (an SBR payload that fails leaves valid "SBR off" records and the unit decodes,
        // as in the reference).  The reference returns an error and writes no samples, aac_decode_frame :2065-2068; here
        // the stream's slot in the batch still runs, so it gets records that are safe to decode -- silence, no tools,
        // no SBR payload -- and submit() parks its state rows around the launches: the stream is left as it was before
        // the unit and its PCM of this tick is zero.  (What was half written by the failed parse, and the SBR / PS
        // records of the tick that used this buffer set last, must not reach the kernels.)
        // One thing does move, as in the reference: where its element decoders had drawn noise or stepped predictors
        // before they refused the unit, the parser has left tools records that do exactly that much
        // (HEAAC_REFUSED_RUN_TOOLS, heaac_parse.h) -- they run, and only the decoder's state rows are parked.
        const int core_failed = r < 0 && fi.channels == 0;
        const int run_tools = core_failed && (fi.refused & HEAAC_REFUSED_RUN_TOOLS);
        s->failed[i] = (unsigned char)(core_failed ? 1 + run_tools : 0);
        if (core_failed) {
            memset(s->h_ics + i * p->ncore, 0, p->ncore * sizeof(HeaacIcs));
            if (!run_tools) {
                memset(s->h_coeffs + i * (size_t)p->ncore * 1024, 0, (size_t)p->ncore * 4096);
                memset(&s->h_tools[i], 0, sizeof(HeaacToolsFrame));
            }
            if (p->he) {
                // the record of "no payload" from a COPY of the stream's SBR state (the call moves kx / m along)
                void *tmp = alloca(heaac_sbr_stream_bytes());
                memcpy(tmp, sst_i, heaac_sbr_stream_bytes());
                heaac_sbr_no_payload((HeaacSbrStream *)tmp, p->ncore, &s->h_sbr[i], s->h_ps ? &s->h_ps[i] : NULL);
            }
        }
    }
}

static void *worker(void *arg)
{
    WorkerArg *a = (WorkerArg *)arg;
    HeaacPipeline *p = a->p;
    const int w = a->w;
    free(a);
    unsigned long seen = 0;
    pthread_mutex_lock(&p->mu);
    for (;;) {
        while (p->generation == seen && !p->quit) pthread_cond_wait(&p->cv_go, &p->mu);
        if (p->quit) break;
        seen = p->generation;
        pthread_mutex_unlock(&p->mu);
        parse_slice(p, w);
        pthread_mutex_lock(&p->mu);
        if (--p->pending == 0) pthread_cond_signal(&p->cv_done);
    }
    pthread_mutex_unlock(&p->mu);
    return NULL;
}

// Parser threads when the caller does not say: the CPUs this process may actually use.  Inside a container the CPU
// bandwidth quota (cgroup v2 cpu.max) can be far below the online count; threads beyond about twice the quota only
// get throttled (measured on the GPU box: 256 online, quota 16: 32 threads parse a tick in 6 ms, 256 in 18 ms
// once the copy engines' helper threads compete).
static int default_threads(void)
{
    long online = sysconf(_SC_NPROCESSORS_ONLN);
    if (online < 1) online = 1;
    FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (f) {
        long long quota = 0, period = 0;
        if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) {
            const long cap = (long)((2 * quota + period - 1) / period);
            if (cap >= 1 && cap < online) online = cap;
        }
        fclose(f);
    }
    return (int)online;
}

static int pinned(void **p, size_t bytes) { return hipHostMalloc(p, bytes, hipHostMallocDefault) == hipSuccess; }
static int devmem(void **p, size_t bytes) { return hipMalloc(p, bytes) == hipSuccess; }

extern "C" void heaac_pipeline_destroy(HeaacPipeline *p)
{
    if (!p) return;
    if (p->tid) {
        pthread_mutex_lock(&p->mu);
        p->quit = 1;
        pthread_cond_broadcast(&p->cv_go);
        pthread_mutex_unlock(&p->mu);
        for (int t = 1; t < p->threads; t++) if (p->tid[t]) pthread_join(p->tid[t], NULL);
        free(p->tid);
        pthread_cond_destroy(&p->cv_go); pthread_cond_destroy(&p->cv_done); pthread_mutex_destroy(&p->mu);
    }
    if (p->in) (void)hipStreamSynchronize(p->in);
    if (p->run) (void)hipStreamSynchronize(p->run);
    if (p->out) (void)hipStreamSynchronize(p->out);
    for (int k = 0; k < PL_DEPTH; k++) {
        Set *s = &p->set[k];
        void *h[] = { s->h_coeffs, s->h_ics, s->h_tools, s->h_sbr, s->h_ps, s->h_pcm, s->h_list };
        void *d[] = { s->d_coeffs, s->d_ics, s->d_tools, s->d_sbr, s->d_ps, s->d_pcm, s->d_list };
        for (void *x : h) if (x) (void)hipHostFree(x);
        for (void *x : d) if (x) (void)hipFree(x);
        hipEvent_t ev[] = { s->in_start, s->in_done, s->run_done, s->out_done };
        for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
        free(s->failed);
    }
    if (p->d_state) (void)hipFree(p->d_state);
    if (p->d_rng) (void)hipFree(p->d_rng);
    if (p->d_pred) (void)hipFree(p->d_pred);
    if (p->d_hdr) (void)hipFree(p->d_hdr);
    if (p->d_park_state) (void)hipFree(p->d_park_state);
    if (p->d_park_rng) (void)hipFree(p->d_park_rng);
    if (p->d_park_pred) (void)hipFree(p->d_park_pred);
    if (p->in) (void)hipStreamDestroy(p->in);
    if (p->run) (void)hipStreamDestroy(p->run);
    if (p->out) (void)hipStreamDestroy(p->out);
    heaac_sbr_table_destroy(p->tab);
    free(p->ast); free(p->sst);
    heaac_device_destroy(p->dev);
    free(p);
}

extern "C" int heaac_pipeline_create(HeaacPipeline **out, const HeaacAacConfig *aac, int he_cfg, size_t n, int threads)
{
    if (!out) return HEAAC_ERR_ARG;
    *out = NULL;
    const bool lc = he_cfg == HEAAC_CFG_LC_MONO || he_cfg == HEAAC_CFG_LC_STEREO;
    if (!aac || !n || (!lc && he_cfg != HEAAC_CFG_HEV2 && he_cfg != HEAAC_CFG_HEV1 && he_cfg != HEAAC_CFG_HEV1_MONO) ||
        aac->sampling_index < 0 || aac->sampling_index > 12)
        return HEAAC_ERR_ARG;
    HeaacPipeline *p = (HeaacPipeline *)calloc(1, sizeof(*p));
    if (!p) return HEAAC_ERR_NOMEM;
    p->aac = *aac;
    p->he_cfg = he_cfg;
    p->he = !lc;
    // the configuration's two sample rates decide between 2048 samples at twice the core rate and "downsampled SBR"
    const int mode = lc ? 0 : heaac_sbr_output_mode(aac);
    if (mode < 0) { free(p); return HEAAC_ERR_ARG; }
    p->downsampled = mode;
    p->out_len = lc || mode ? 1024 : 2048;
    p->ncore = (he_cfg == HEAAC_CFG_HEV1 || he_cfg == HEAAC_CFG_LC_STEREO) ? 2 : 1;
    p->nout = (he_cfg == HEAAC_CFG_HEV1_MONO || he_cfg == HEAAC_CFG_LC_MONO) ? 1 : 2;
    p->words = he_cfg == HEAAC_CFG_HEV1 ? HEAAC_STATE_WORDS_HEV1 : he_cfg == HEAAC_CFG_HEV2 ? HEAAC_STATE_WORDS_HEV2 :
               he_cfg == HEAAC_CFG_HEV1_MONO ? HEAAC_STATE_WORDS_HEV1_MONO :
               he_cfg == HEAAC_CFG_LC_STEREO ? HEAAC_STATE_WORDS_LC_STEREO : HEAAC_STATE_WORDS_LC_MONO;
    p->n = n;
    int rc = heaac_device_create(&p->dev, n);
    if (rc != HEAAC_OK) { free(p); return rc; }
    bool ok = hipStreamCreateWithFlags(&p->in, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&p->run, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&p->out, hipStreamNonBlocking) == hipSuccess;
    const int with_ps = he_cfg == HEAAC_CFG_HEV2;
    for (int k = 0; k < PL_DEPTH && ok; k++) {
        Set *s = &p->set[k];
        const size_t nc = n * (size_t)p->ncore;
        ok = pinned((void **)&s->h_coeffs, nc * 4096) && pinned((void **)&s->h_ics, nc * sizeof(HeaacIcs)) &&
             pinned((void **)&s->h_tools, n * sizeof(HeaacToolsFrame)) &&
             (!p->he || pinned((void **)&s->h_sbr, n * sizeof(HeaacSbrFrame))) &&
             (!with_ps || pinned((void **)&s->h_ps, n * sizeof(HeaacPsFrame))) &&
             pinned((void **)&s->h_pcm, n * (size_t)p->nout * p->out_len * 2) &&
             devmem((void **)&s->d_coeffs, nc * 4096) && devmem((void **)&s->d_ics, nc * sizeof(HeaacIcs)) &&
             devmem((void **)&s->d_tools, n * sizeof(HeaacToolsFrame)) &&
             (!p->he || devmem((void **)&s->d_sbr, n * sizeof(HeaacSbrFrame))) &&
             (!with_ps || devmem((void **)&s->d_ps, n * sizeof(HeaacPsFrame))) &&
             devmem((void **)&s->d_pcm, n * (size_t)p->nout * p->out_len * 2) &&
             (s->failed = (unsigned char *)calloc(n, 1)) != NULL &&
             pinned((void **)&s->h_list, 4 * n * sizeof(unsigned)) && devmem((void **)&s->d_list, 4 * n * sizeof(unsigned)) &&
             hipEventCreate(&s->in_start) == hipSuccess && hipEventCreate(&s->in_done) == hipSuccess &&
             hipEventCreate(&s->run_done) == hipSuccess && hipEventCreate(&s->out_done) == hipSuccess;
        if (ok) {
            memset(s->h_coeffs, 0, nc * 4096); memset(s->h_ics, 0, nc * sizeof(HeaacIcs));
            memset(s->h_tools, 0, n * sizeof(HeaacToolsFrame));
            if (p->he) memset(s->h_sbr, 0, n * sizeof(HeaacSbrFrame));
            if (with_ps) memset(s->h_ps, 0, n * sizeof(HeaacPsFrame));
            ok = hipMemset(s->d_tools, 0, n * sizeof(HeaacToolsFrame)) == hipSuccess;
        }
    }
    ok = ok && devmem((void **)&p->d_state, n * p->words * 4) && devmem((void **)&p->d_rng, n * 4) &&
         devmem((void **)&p->d_hdr, PL_MAX_HDRS * sizeof(HeaacSbrHeader)) &&
         hipMemset(p->d_state, 0, n * p->words * 4) == hipSuccess;
    if (ok && aac->object_type == HEAAC_AOT_AAC_MAIN) {
        // reset_predict_state (aacdec.c:507-515) for every predictor of every channel
        const size_t np = n * (size_t)p->ncore * HEAAC_MAX_PREDICTORS;
        HeaacPredictorState *ps = (HeaacPredictorState *)calloc(np, sizeof(*ps));
        ok = ps != NULL && devmem((void **)&p->d_pred, np * sizeof(*ps));
        if (ok) {
            for (size_t i = 0; i < np; i++) ps[i].var0 = ps[i].var1 = 1.0f;
            ok = hipMemcpy(p->d_pred, ps, np * sizeof(*ps), hipMemcpyHostToDevice) == hipSuccess;
        }
        free(ps);
    }
    if (ok) {
        int32_t *seed = (int32_t *)malloc(n * 4);
        ok = seed != NULL;
        if (ok) {
            for (size_t i = 0; i < n; i++) seed[i] = 0x1f2e3d4c;       // ac->random_state, aacdec.c:558
            ok = hipMemcpy(p->d_rng, seed, n * 4, hipMemcpyHostToDevice) == hipSuccess;
            free(seed);
        }
    }
    p->tab = heaac_sbr_table_create(PL_MAX_HDRS);
    p->ast = (HeaacAacStream *)calloc(n, sizeof(HeaacAacStream));
    p->sst = (HeaacSbrStream *)malloc(n * heaac_sbr_stream_bytes());
    ok = ok && p->tab && p->ast && p->sst;
    if (ok) {
        heaac_sbr_stream_init(p->sst, n);
        // the null header (table entry 0) is what frames before their stream's first header point at
        ok = hipMemcpy(p->d_hdr, heaac_sbr_table_data(p->tab), sizeof(HeaacSbrHeader), hipMemcpyHostToDevice) == hipSuccess;
        p->hdr_uploaded = 1;
    }
    if (ok) {
        if (threads <= 0) threads = default_threads();
        if (threads < 1) threads = 1;
        if (threads > 256) threads = 256;
        if ((size_t)threads > n) threads = (int)n;
        p->threads = threads;
        pthread_mutex_init(&p->mu, NULL);
        pthread_cond_init(&p->cv_go, NULL);
        pthread_cond_init(&p->cv_done, NULL);
        p->tid = (pthread_t *)calloc(threads, sizeof(pthread_t));
        ok = p->tid != NULL;
        for (int t = 1; t < threads && ok; t++) {          // slice 0 is parsed by the submitting thread
            WorkerArg *a = (WorkerArg *)malloc(sizeof(*a));
            if (!a) { ok = false; break; }
            a->p = p; a->w = t;
            if (pthread_create(&p->tid[t], NULL, worker, a) != 0) { free(a); p->tid[t] = 0; p->threads = t; break; }
        }
    }
    if (!ok) { heaac_pipeline_destroy(p); return HEAAC_ERR_NOMEM; }
    *out = p;
    return HEAAC_OK;
}

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

#define HIP_OK(x) do { if ((x) != hipSuccess) return HEAAC_ERR_HIP; } while (0)

extern "C" int heaac_pipeline_submit(HeaacPipeline *p, const uint8_t *const *au, const int *size, int *status)
{
    if (!p || !au || !size) return HEAAC_ERR_ARG;
    if (p->submitted - p->collected >= PL_DEPTH) return HEAAC_ERR_ARG;
    Set *s = &p->set[p->submitted % PL_DEPTH];
    // the pinned inputs of this set are free once the H2D of the tick that used it last has finished
    if (s->used) HIP_OK(hipEventSynchronize(s->in_done));
    const double t0 = now_ms();
    pthread_mutex_lock(&p->mu);
    p->job_au = au; p->job_size = size; p->job_status = status; p->job_set = s;
    p->pending = p->threads - 1;
    p->generation++;
    pthread_cond_broadcast(&p->cv_go);
    pthread_mutex_unlock(&p->mu);
    parse_slice(p, 0);
    pthread_mutex_lock(&p->mu);
    while (p->pending > 0) pthread_cond_wait(&p->cv_done, &p->mu);
    pthread_mutex_unlock(&p->mu);
    s->parse_ms = (float)(now_ms() - t0);

    const size_t n = p->n, nc = n * (size_t)p->ncore;
    // new SBR headers of this tick (the table's storage never moves)
    const size_t have = heaac_sbr_table_count(p->tab);
    if (have > PL_MAX_HDRS) return HEAAC_ERR_ARG;
    // H2D
    if (s->used) HIP_OK(hipStreamWaitEvent(p->in, s->run_done, 0));
    HIP_OK(hipEventRecord(s->in_start, p->in));
    if (have > p->hdr_uploaded) {
        HIP_OK(hipMemcpyAsync(p->d_hdr + p->hdr_uploaded, heaac_sbr_table_data(p->tab) + p->hdr_uploaded,
                              (have - p->hdr_uploaded) * sizeof(HeaacSbrHeader), hipMemcpyHostToDevice, p->in));
        p->hdr_uploaded = have;
    }
    HIP_OK(hipMemcpyAsync(s->d_coeffs, s->h_coeffs, nc * 4096, hipMemcpyHostToDevice, p->in));
    HIP_OK(hipMemcpyAsync(s->d_ics, s->h_ics, nc * sizeof(HeaacIcs), hipMemcpyHostToDevice, p->in));
    if (p->ncore == 1) {
        // a mono stream uses channel 0 of the tools record only: the second channel's 3.5 KB stay on the host
        // (the device copies were zeroed once and the kernel never reads them for one channel)
        HIP_OK(hipMemcpy2DAsync(s->d_tools, sizeof(HeaacToolsFrame), s->h_tools, sizeof(HeaacToolsFrame),
                                offsetof(HeaacToolsFrame, ch) + sizeof(HeaacToolsChannel), n, hipMemcpyHostToDevice, p->in));
    } else {
        HIP_OK(hipMemcpyAsync(s->d_tools, s->h_tools, n * sizeof(HeaacToolsFrame), hipMemcpyHostToDevice, p->in));
    }
    if (p->he) HIP_OK(hipMemcpyAsync(s->d_sbr, s->h_sbr, n * sizeof(HeaacSbrFrame), hipMemcpyHostToDevice, p->in));
    if (s->d_ps) HIP_OK(hipMemcpyAsync(s->d_ps, s->h_ps, n * sizeof(HeaacPsFrame), hipMemcpyHostToDevice, p->in));
    HIP_OK(hipEventRecord(s->in_done, p->in));
    // GPU
    HIP_OK(hipStreamWaitEvent(p->run, s->in_done, 0));
    if (s->used) HIP_OK(hipStreamWaitEvent(p->run, s->out_done, 0));
    // streams whose unit failed: park their state rows (state, noise generator, predictors) before the launches ...
    size_t n_failed = 0;
    for (size_t i = 0; i < n; i++) n_failed += s->failed[i] != 0;
    const size_t pred_row = (size_t)p->ncore * HEAAC_MAX_PREDICTORS;
    if (n_failed > p->park_rows) {
        // (grown outside the streams' order: nothing of the old area is in flight once `run` has drained)
        HIP_OK(hipStreamSynchronize(p->run));
        if (p->d_park_state) (void)hipFree(p->d_park_state);
        if (p->d_park_rng) (void)hipFree(p->d_park_rng);
        if (p->d_park_pred) (void)hipFree(p->d_park_pred);
        p->d_park_state = NULL; p->d_park_rng = NULL; p->d_park_pred = NULL; p->park_rows = 0;
        size_t rows = 64;
        while (rows < n_failed) rows *= 2;
        if (rows > n) rows = n;
        if (!devmem((void **)&p->d_park_state, rows * p->words * 4) || !devmem((void **)&p->d_park_rng, rows * 4) ||
            (p->d_pred && !devmem((void **)&p->d_park_pred, rows * pred_row * sizeof(HeaacPredictorState))))
            return HEAAC_ERR_NOMEM;
        p->park_rows = rows;
    }
    // two lists of (stream, parking row): every failed stream (its DSP state, its PCM row), and those of them whose
    // generator and predictors stay put as well (failed == 1; 2: the tools' side of the stream moves on)
    unsigned n_all = 0, n_full = 0;
    unsigned *list_all = s->h_list, *list_full = s->h_list + 2 * n;
    if (n_failed) {
        for (size_t i = 0; i < n; i++) {
            if (!s->failed[i]) continue;
            if (s->failed[i] == 1) { list_full[2 * n_full] = (unsigned)i; list_full[2 * n_full + 1] = n_all; n_full++; }
            list_all[2 * n_all] = (unsigned)i; list_all[2 * n_all + 1] = n_all; n_all++;
        }
        HIP_OK(hipMemcpyAsync(s->d_list, list_all, 2 * n_all * sizeof(unsigned), hipMemcpyHostToDevice, p->run));
        if (n_full)
            HIP_OK(hipMemcpyAsync(s->d_list + 2 * n, list_full, 2 * n_full * sizeof(unsigned), hipMemcpyHostToDevice, p->run));
        hipLaunchKernelGGL(k_rows, dim3(n_all), dim3(256), 0, p->run, s->d_list, (unsigned *)p->d_state, (unsigned *)p->d_park_state,
                           (unsigned long long)p->words, 0);
        if (n_full) {
            hipLaunchKernelGGL(k_rows, dim3(n_full), dim3(64), 0, p->run, s->d_list + 2 * n, (unsigned *)p->d_rng, (unsigned *)p->d_park_rng, 1ull, 0);
            if (p->d_pred)
                hipLaunchKernelGGL(k_rows, dim3(n_full), dim3(256), 0, p->run, s->d_list + 2 * n, (unsigned *)p->d_pred, (unsigned *)p->d_park_pred,
                                   (unsigned long long)(pred_row * sizeof(HeaacPredictorState) / 4), 0);
        }
        HIP_OK(hipGetLastError());
    }
    int rc = heaac_spectral_tools_batch(p->dev, p->ncore, s->d_coeffs, s->d_tools, p->d_rng, p->d_rng, p->d_pred, p->d_pred, n,
                                        (void *)p->run);
    if (rc == HEAAC_OK)
        rc = p->he ? heaac_he_decode_batch_ex(p->dev, p->he_cfg, p->downsampled ? HEAAC_HE_DOWNSAMPLED : 0, s->d_coeffs, s->d_ics,
                                              s->d_sbr, p->d_hdr, PL_MAX_HDRS, s->d_ps,
                                              p->d_state, p->d_state, s->d_pcm, HEAAC_PCM_S16_INTERLEAVED, n, (void *)p->run)
                   : heaac_lc_decode_batch(p->dev, p->ncore, s->d_coeffs, s->d_ics, p->d_state, p->d_state, s->d_pcm,
                                           HEAAC_PCM_S16_INTERLEAVED, n, (void *)p->run);
    if (rc != HEAAC_OK) return rc;
    // ... and put them back, with silence where the decode wrote
    if (n_failed) {
        const size_t pcm_row = (size_t)p->nout * p->out_len;           // int16: an even count, so whole 32-bit words
        hipLaunchKernelGGL(k_rows, dim3(n_all), dim3(256), 0, p->run, s->d_list, (unsigned *)p->d_state, (unsigned *)p->d_park_state,
                           (unsigned long long)p->words, 1);
        hipLaunchKernelGGL(k_rows, dim3(n_all), dim3(256), 0, p->run, s->d_list, (unsigned *)s->d_pcm, (unsigned *)nullptr,
                           (unsigned long long)(pcm_row / 2), 2);
        if (n_full) {
            hipLaunchKernelGGL(k_rows, dim3(n_full), dim3(64), 0, p->run, s->d_list + 2 * n, (unsigned *)p->d_rng, (unsigned *)p->d_park_rng, 1ull, 1);
            if (p->d_pred)
                hipLaunchKernelGGL(k_rows, dim3(n_full), dim3(256), 0, p->run, s->d_list + 2 * n, (unsigned *)p->d_pred, (unsigned *)p->d_park_pred,
                                   (unsigned long long)(pred_row * sizeof(HeaacPredictorState) / 4), 1);
        }
        HIP_OK(hipGetLastError());
    }
    HIP_OK(hipEventRecord(s->run_done, p->run));
    // D2H
    HIP_OK(hipStreamWaitEvent(p->out, s->run_done, 0));
    HIP_OK(hipMemcpyAsync(s->h_pcm, s->d_pcm, n * (size_t)p->nout * p->out_len * 2, hipMemcpyDeviceToHost, p->out));
    HIP_OK(hipEventRecord(s->out_done, p->out));
    s->used = 1;
    p->submitted++;
    return HEAAC_OK;
}

extern "C" int heaac_pipeline_collect(HeaacPipeline *p, const int16_t **pcm)
{
    if (!p || !pcm || p->collected == p->submitted) return HEAAC_ERR_ARG;
    Set *s = &p->set[p->collected % PL_DEPTH];
    HIP_OK(hipEventSynchronize(s->out_done));
    *pcm = s->h_pcm;
    p->last_ms[0] = s->parse_ms;
    (void)hipEventElapsedTime(&p->last_ms[1], s->in_start, s->in_done);
    (void)hipEventElapsedTime(&p->last_ms[2], s->in_done, s->run_done);
    (void)hipEventElapsedTime(&p->last_ms[3], s->run_done, s->out_done);
    p->collected++;
    return HEAAC_OK;
}

extern "C" void heaac_pipeline_timing(const HeaacPipeline *p, float ms[4])
{
    for (int k = 0; k < 4; k++) ms[k] = p ? p->last_ms[k] : 0.0f;
}
