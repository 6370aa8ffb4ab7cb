// pipeline_layout.hip -- include/heaac_pipeline.h, second half: n streams of ONE multi-element layout (3.0 ... 7.1, a
// program config layout) advance in lock step, access units in host memory in, interleaved int16 PCM in host memory
// out.  What codec_layout.hip does for one stream with batches of one -- aac_decode_frame's element loop
// (aacdec.c:1999-2076), spectral_to_sample per element (:1903-1933), float_to_int16_interleave over output_data[]
// (:2096-2097) -- done for all streams at once: the parsed records are laid out ELEMENT-major
// ([element][stream]), so every element of the layout is one batched tools call and one batched decode call over the
// n streams, and one interleave launch writes [n][len][channels].
//   * one noise generator per stream, run through the elements in bitstream order: the streams of a pipeline must
//     share that order (encoders emit one order; the first good unit sets it).  A stream that deviates, leaves an
//     element out or fails to parse gets silence for the tick and keeps its decoder state (as heaac_pipeline does);
//   * SBR per element (explicit signalling: m4ac.sbr = 1), "pure upsampling" where an element has no payload;
//   * layouts whose program config element names coupling channel elements are not taken here (one
//     heaac_codec_decode context per such stream): HEAAC_ERR_ARG at create.
// Two buffer sets rotate: the host parses tick t + 1 while tick t is on the link and on the GPU.
#include <hip/hip_runtime.h>
#include <alloca.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "heaac_pipeline.h"
#include "codec_layout.h"        // heaac_sbr_output_mode

#define LP_DEPTH 2
#define LP_MAX_HDRS 4096

struct LpElem {
    int channels, cfg, words;
    int out, first_out;             // planes the element gives (2 for an SCE with Parametric Stereo) and where they stand
    // persistent, device
    float *d_state;                 // [n][words]
    HeaacPredictorState *d_pred;    // [n][channels][672] (AAC-Main) or NULL
    float *d_f32;                   // [n][channels][len]: the element's planes of the tick in work
    HeaacSbrStream *sst;            // [n] host: the element's SBR reader state per stream
};

// The coupling channel elements of a layout (slots 0 .. K-1 of the layout's list): individual channel streams of
// their own, batched per slot like an output element
struct LpCoupled {
    int K, words;                                   // words: 512 (the overlap), behind SBR a mono element's whole state
    float *d_state[HEAAC_MAX_CCE];                  // [n][words]: state of a coupling channel that couples AFTER_IMDCT
    HeaacPredictorState *d_pred[HEAAC_MAX_CCE];     // [n][672] (AAC-Main)
    float *d_ret[HEAAC_MAX_CCE];                    // [n][len]: the coupling channel's own output
    float *d_state_tmp;                             // [n][words]
    float *d_plane;                                 // [n][len]: one channel of a two-channel element, gathered for the coupling op
    HeaacSbrStream *sst[HEAAC_MAX_CCE];             // [n] host: a coupling channel's own SBR reader (che->sbr)
    unsigned char *seen;                            // [n][K]: an earlier unit of the stream carried the element
};

struct LpSet {
    // coupling elements: per output element the records that land there, per slot the channel's own records
    HeaacCceFrame *h_cce[HEAAC_MAX_ELEMENTS], *d_cce[HEAAC_MAX_ELEMENTS];   // [n][K]
    float *h_ccoef[HEAAC_MAX_CCE], *d_ccoef[HEAAC_MAX_CCE];                 // [n][1024]
    HeaacIcs *h_cics[HEAAC_MAX_CCE], *d_cics[HEAAC_MAX_CCE];
    HeaacToolsFrame *h_ctools[HEAAC_MAX_CCE], *d_ctools[HEAAC_MAX_CCE];
    HeaacSbrFrame *h_csbr[HEAAC_MAX_CCE], *d_csbr[HEAAC_MAX_CCE];
    float *d_ccoef_fm;                              // [n][K][1024]: the spectra frame-major, as the tools' POST half reads them
    HeaacCoupling *h_gain, *d_gain;                 // [ne][K][HEAAC_MAX_CCE_LINKS][n * len / 1024]: AFTER_IMDCT gains, call by call
    unsigned char *cpos;                            // [n][K][3]: present, outputs_before, seq of each coupling element
    unsigned *h_list, *d_list;                      // [2][2 n] (stream, parking row) pairs of the failed streams; then K x [2 n]:
                                                    // per coupling slot the streams whose element couples AFTER_IMDCT
    // per element, [n] each; pinned host / device
    float *h_coeffs[HEAAC_MAX_ELEMENTS], *d_coeffs[HEAAC_MAX_ELEMENTS];
    HeaacIcs *h_ics[HEAAC_MAX_ELEMENTS], *d_ics[HEAAC_MAX_ELEMENTS];
    HeaacToolsFrame *h_tools[HEAAC_MAX_ELEMENTS], *d_tools[HEAAC_MAX_ELEMENTS];
    HeaacSbrFrame *h_sbr[HEAAC_MAX_ELEMENTS], *d_sbr[HEAAC_MAX_ELEMENTS];
    HeaacPsFrame *h_ps[HEAAC_MAX_ELEMENTS], *d_ps[HEAAC_MAX_ELEMENTS];     // single channel elements with Parametric Stereo only
    int16_t *h_pcm, *d_pcm;
    unsigned char *failed;          // [n]
    unsigned char *seq;             // [n][ne] bitstream position of each element
    hipEvent_t done;
    int used;
};

struct HeaacLayoutPipeline {
    HeaacAacConfig aac;
    HeaacAacLayout layout;
    int ne, he, downsampled, len, main_profile;
    int out_channels;               // the layout's channels, plus one per SCE with Parametric Stereo
    size_t n;
    HeaacDevice *dev;
    hipStream_t run;
    LpElem e[HEAAC_MAX_ELEMENTS];
    LpSet set[LP_DEPTH];
    int32_t *d_rng;
    HeaacSbrHeader *d_hdr; size_t hdr_uploaded;
    HeaacSbrHeaderTable *tab;
    HeaacAacStream *ast;            // [n][ne]
    HeaacAacLayout *lay;            // [n]: every stream's own tag map
    int have_order; unsigned char order[HEAAC_MAX_ELEMENTS];     // order[seq] = element at that bitstream position
    LpCoupled *cpl;                 // layouts whose program config element names coupling elements (AAC-LC / Main)
    // parked state of the streams whose unit failed
    float *d_park; size_t park_bytes, park_rows;
    unsigned long submitted, collected;
    // pool
    int threads;
    pthread_t *tid;
    pthread_mutex_t mu;
    pthread_cond_t cv_go, cv_done;
    unsigned long generation;
    int pending, quit;
    const uint8_t *const *job_au; const int *job_size; int *job_status; LpSet *job_set;
};

struct LpWorkerArg { HeaacLayoutPipeline *p; int w; };

static inline HeaacSbrStream *sst_at(HeaacSbrStream *base, size_t i)
{
    return (HeaacSbrStream *)((char *)base + i * heaac_sbr_stream_bytes());
}

// neutral records of one stream: silence, no tools, no SBR payload (from a COPY of the SBR reader state).
// keep_tools: the tools records and spectra are already what a refused unit leaves for the spectral tools.
static void neutral(HeaacLayoutPipeline *p, LpSet *s, size_t i, bool keep_tools = false)
{
    if (p->cpl) {
        const int K = p->cpl->K;
        for (int e = 0; e < p->ne; e++) memset(s->h_cce[e] + i * K, 0, (size_t)K * sizeof(HeaacCceFrame));
        for (int k = 0; k < K; k++) {
            memset(s->h_ccoef[k] + i * 1024, 0, 4096);
            memset(&s->h_cics[k][i], 0, sizeof(HeaacIcs));
            memset(&s->h_ctools[k][i], 0, sizeof(HeaacToolsFrame));
            if (p->he) {
                void *tmp = alloca(heaac_sbr_stream_bytes());
                memcpy(tmp, sst_at(p->cpl->sst[k], i), heaac_sbr_stream_bytes());
                heaac_sbr_no_payload((HeaacSbrStream *)tmp, 1, &s->h_csbr[k][i], NULL);
            }
        }
        memset(s->cpos + i * K * 3, 0, (size_t)K * 3);
    }
    for (int k = 0; k < p->ne; k++) {
        const int ch = p->e[k].channels;
        memset(s->h_ics[k] + i * ch, 0, ch * sizeof(HeaacIcs));
        if (!keep_tools) {
            memset(s->h_coeffs[k] + i * (size_t)ch * 1024, 0, (size_t)ch * 4096);
            memset(&s->h_tools[k][i], 0, sizeof(HeaacToolsFrame));
        }
        if (p->he) {
            void *tmp = alloca(heaac_sbr_stream_bytes());
            memcpy(tmp, sst_at(p->e[k].sst, i), heaac_sbr_stream_bytes());
            heaac_sbr_no_payload((HeaacSbrStream *)tmp, ch, &s->h_sbr[k][i], s->h_ps[k] ? &s->h_ps[k][i] : NULL);
        }
    }
}

static void lp_parse_slice(HeaacLayoutPipeline *p, int w)
{
    const size_t lo = p->n * (size_t)w / (size_t)p->threads, hi = p->n * (size_t)(w + 1) / (size_t)p->threads;
    LpSet *s = p->job_set;
    const int ne = p->ne;
    float *tc = (float *)malloc((size_t)ne * 2048 * sizeof(float));
    HeaacIcs *ti = (HeaacIcs *)malloc((size_t)ne * 2 * sizeof(HeaacIcs));
    HeaacToolsFrame *tt = (HeaacToolsFrame *)malloc((size_t)ne * sizeof(HeaacToolsFrame));
    HeaacAacElementInfo te[HEAAC_MAX_ELEMENTS];
    HeaacAacStream st[HEAAC_MAX_ELEMENTS];
    // the parser's coupling output (rows of HEAAC_MAX_CCE slots per output element)
    const int K = p->cpl ? p->cpl->K : 0;
    HeaacCceOut co = { NULL, NULL, NULL, NULL, NULL };
    HeaacAacElementInfo ce[HEAAC_MAX_CCE];
    if (K) {
        co.cce = (HeaacCceFrame *)malloc((size_t)ne * HEAAC_MAX_CCE * sizeof(HeaacCceFrame));
        co.coeffs = (float *)malloc(HEAAC_MAX_CCE * 4096);
        co.ics = (HeaacIcs *)malloc(HEAAC_MAX_CCE * sizeof(HeaacIcs));
        co.tools = (HeaacToolsFrame *)malloc(HEAAC_MAX_CCE * sizeof(HeaacToolsFrame));
        co.elem = ce;
        if (!co.cce || !co.coeffs || !co.ics || !co.tools) { free(tc); tc = NULL; }
    }
    for (size_t i = lo; i < hi; i++) {
        int r = HEAAC_PARSE_ERR_ARG;
        HeaacAacFrameInfo fi;
        memset(&fi, 0, sizeof(fi));
        if (tc && ti && tt) {
            // the parser works on copies of the window histories until the whole unit has parsed
            for (int k = 0; k < ne; k++) st[k] = p->ast[i * ne + k];
            r = heaac_aac_parse_frame_layout_ex(&p->aac, &p->lay[i], st, p->job_au[i], p->job_size[i], tc, ti, tt, te, K ? &co : NULL, &fi);
            for (int k = 0; r == HEAAC_PARSE_OK && k < ne; k++)
                if (!te[k].present) r = HEAAC_PARSE_ERR_DATA;      // an element of the layout left out (codec_layout.hip: refused)
            // ... or a coupling element an earlier unit of the stream carried (the reference would couple whatever its
            // buffers still hold)
            for (int k = 0; r == HEAAC_PARSE_OK && k < K; k++)
                if (p->cpl->seen[i * K + k] && !co.cce[k].present) r = HEAAC_PARSE_ERR_DATA;
        }
        if (p->job_status) p->job_status[i] = r;
        s->failed[i] = (unsigned char)(r != HEAAC_PARSE_OK);
        if (r != HEAAC_PARSE_OK) {
            // Where the refusal is the reference's own, what its element decoders had done by then stays done
            // (heaac_parse.h, HEAAC_REFUSED_*): the window histories the parser has moved, and -- through the records it
            // left for the elements marked present -- the noise generator and the predictors (failed = 2: the tools
            // run on them, submit() parks the DSP state rows only).
            if (r < 0 && tc && (fi.refused & HEAAC_REFUSED_AS_REFERENCE))
                for (int k = 0; k < ne; k++) p->ast[i * ne + k] = st[k];
            if (r < 0 && tc && (fi.refused & HEAAC_REFUSED_RUN_TOOLS)) {
                s->failed[i] = 2;
                for (int k = 0; k < ne; k++) {
                    const int ch = p->e[k].channels;
                    s->seq[i * ne + k] = te[k].present ? te[k].seq : 0xff;
                    if (te[k].present) {
                        memcpy(s->h_coeffs[k] + i * (size_t)ch * 1024, tc + (size_t)k * 2048, (size_t)ch * 4096);
                        s->h_tools[k][i] = tt[k];
                    } else {
                        memset(s->h_coeffs[k] + i * (size_t)ch * 1024, 0, (size_t)ch * 4096);
                        memset(&s->h_tools[k][i], 0, sizeof(HeaacToolsFrame));
                    }
                }
            }
            neutral(p, s, i, s->failed[i] == 2);
            continue;
        }
        for (int k = 0; k < K; k++) {
            const HeaacCceFrame &c0 = co.cce[k];            // (the same element in every output slot's row)
            unsigned char *cp = s->cpos + (i * K + k) * 3;
            cp[0] = c0.present; cp[1] = c0.outputs_before; cp[2] = c0.seq;
            if (c0.present) p->cpl->seen[i * K + k] = 1;
            for (int e = 0; e < ne; e++) {
                HeaacCceFrame &o = s->h_cce[e][i * K + k];
                o = co.cce[e * HEAAC_MAX_CCE + k];
                // "Dependent coupling is not supported together with LTP" (apply_dependent_coupling :1822-1826 returns)
                if (p->aac.object_type == 4 && o.coupling_point != HEAAC_CC_AFTER_IMDCT) o.n_links = 0;
            }
            if (p->he) {
                // A coupling channel's own SBR (it goes through ff_sbr_apply when it couples AFTER_IMDCT, aacdec.c:1920-1927).
                // A payload behind an element that couples in the spectrum is read all the same; without one the
                // reader's state only moves where the channel is transformed.
                HeaacSbrStream *cs = sst_at(p->cpl->sst[k], i);
                const bool after = c0.present && c0.coupling_point == HEAAC_CC_AFTER_IMDCT;
                if (c0.present && ce[k].sbr_payload_bit >= 0) {
                    (void)heaac_sbr_parse_payload(cs, p->tab, p->aac.sample_rate, p->job_au[i], p->job_size[i], ce[k].sbr_payload_bit,
                                                  ce[k].sbr_payload_bytes, ce[k].sbr_crc, 1, ce[k].sbr_misplaced ? HEAAC_SBR_MISPLACED : 0,
                                                  &s->h_csbr[k][i], NULL, NULL);
                } else if (after) {
                    heaac_sbr_no_payload(cs, 1, &s->h_csbr[k][i], NULL);
                } else {
                    void *tmp = alloca(heaac_sbr_stream_bytes());
                    memcpy(tmp, cs, heaac_sbr_stream_bytes());
                    heaac_sbr_no_payload((HeaacSbrStream *)tmp, 1, &s->h_csbr[k][i], NULL);
                }
            }
            if (c0.present) {
                memcpy(s->h_ccoef[k] + i * 1024, co.coeffs + k * 1024, 4096);
                s->h_cics[k][i] = co.ics[k];
                s->h_ctools[k][i] = co.tools[k];
            } else {
                memset(s->h_ccoef[k] + i * 1024, 0, 4096);
                memset(&s->h_cics[k][i], 0, sizeof(HeaacIcs));
                memset(&s->h_ctools[k][i], 0, sizeof(HeaacToolsFrame));
            }
        }
        for (int k = 0; k < ne; k++) {
            const int ch = p->e[k].channels;
            p->ast[i * ne + k] = st[k];
            s->seq[i * ne + k] = te[k].seq;
            memcpy(s->h_coeffs[k] + i * (size_t)ch * 1024, tc + (size_t)k * 2048, (size_t)ch * 4096);
            memcpy(s->h_ics[k] + i * ch, ti + k * 2, ch * sizeof(HeaacIcs));
            s->h_tools[k][i] = tt[k];
            if (p->he) {
                // a failed payload leaves its degraded record (start = 0) and the decode goes on, as ff_sbr_apply does
                HeaacPsFrame *ps = s->h_ps[k] ? &s->h_ps[k][i] : NULL;
                if (te[k].sbr_payload_bit >= 0)
                    (void)heaac_sbr_parse_payload(sst_at(p->e[k].sst, i), p->tab, p->aac.sample_rate, p->job_au[i], p->job_size[i],
                                                  te[k].sbr_payload_bit, te[k].sbr_payload_bytes, te[k].sbr_crc, ch,
                                                  (te[k].sbr_misplaced ? HEAAC_SBR_MISPLACED : 0) | (ps ? HEAAC_SBR_ALLOW_PS : 0),
                                                  &s->h_sbr[k][i], ps, NULL);
                else
                    heaac_sbr_no_payload(sst_at(p->e[k].sst, i), ch, &s->h_sbr[k][i], ps);
            }
        }
    }
    free(tc); free(ti); free(tt);
    free(co.cce); free(co.coeffs); free(co.ics); free(co.tools);
}

static void *lp_worker(void *arg)
{
    LpWorkerArg *a = (LpWorkerArg *)arg;
    HeaacLayoutPipeline *p = a->p;
    const int w = a->w;
    free(a);
    unsigned long seen = 0;
    pthread_mutex_lock(&p->mu);
    for (;;) {
        while (p->generation == seen && !p->quit) pthread_cond_wait(&p->cv_go, &p->mu);
        if (p->quit) break;
        seen = p->generation;
        pthread_mutex_unlock(&p->mu);
        lp_parse_slice(p, w);
        pthread_mutex_lock(&p->mu);
        if (--p->pending == 0) pthread_cond_signal(&p->cv_done);
    }
    pthread_mutex_unlock(&p->mu);
    return NULL;
}

static int lp_pinned(void **p, size_t bytes) { return hipHostMalloc(p, bytes, hipHostMallocDefault) == hipSuccess; }
static int lp_devmem(void **p, size_t bytes) { return hipMalloc(p, bytes) == hipSuccess; }

extern "C" void heaac_layout_pipeline_destroy(HeaacLayoutPipeline *p)
{
    if (!p) return;
    if (p->tid) {
        pthread_mutex_lock(&p->mu);
        p->quit = 1;
        pthread_cond_broadcast(&p->cv_go);
        pthread_mutex_unlock(&p->mu);
        for (int t = 1; t < p->threads; t++) if (p->tid[t]) pthread_join(p->tid[t], NULL);
        free(p->tid);
        pthread_mutex_destroy(&p->mu);
        pthread_cond_destroy(&p->cv_go);
        pthread_cond_destroy(&p->cv_done);
    }
    if (p->run) (void)hipStreamSynchronize(p->run);
    for (int q = 0; q < LP_DEPTH; q++) {
        LpSet *s = &p->set[q];
        for (int k = 0; k < HEAAC_MAX_ELEMENTS; k++) {
            void *h[] = { s->h_coeffs[k], s->h_ics[k], s->h_tools[k], s->h_sbr[k], s->h_ps[k] };
            void *d[] = { s->d_coeffs[k], s->d_ics[k], s->d_tools[k], s->d_sbr[k], s->d_ps[k] };
            for (void *x : h) if (x) (void)hipHostFree(x);
            for (void *x : d) if (x) (void)hipFree(x);
        }
        for (int k = 0; k < HEAAC_MAX_ELEMENTS; k++) {
            if (s->h_cce[k]) (void)hipHostFree(s->h_cce[k]);
            if (s->d_cce[k]) (void)hipFree(s->d_cce[k]);
        }
        for (int k = 0; k < HEAAC_MAX_CCE; k++) {
            void *h[] = { s->h_ccoef[k], s->h_cics[k], s->h_ctools[k], s->h_csbr[k] };
            void *d[] = { s->d_ccoef[k], s->d_cics[k], s->d_ctools[k], s->d_csbr[k] };
            for (void *x : h) if (x) (void)hipHostFree(x);
            for (void *x : d) if (x) (void)hipFree(x);
        }
        if (s->h_list) (void)hipHostFree(s->h_list);
        if (s->d_list) (void)hipFree(s->d_list);
        if (s->d_ccoef_fm) (void)hipFree(s->d_ccoef_fm);
        if (s->h_gain) (void)hipHostFree(s->h_gain);
        if (s->d_gain) (void)hipFree(s->d_gain);
        free(s->cpos);
        if (s->h_pcm) (void)hipHostFree(s->h_pcm);
        if (s->d_pcm) (void)hipFree(s->d_pcm);
        if (s->done) (void)hipEventDestroy(s->done);
        free(s->failed); free(s->seq);
    }
    for (int k = 0; k < HEAAC_MAX_ELEMENTS; k++) {
        if (p->e[k].d_state) (void)hipFree(p->e[k].d_state);
        if (p->e[k].d_pred) (void)hipFree(p->e[k].d_pred);
        if (p->e[k].d_f32) (void)hipFree(p->e[k].d_f32);
        free(p->e[k].sst);
    }
    if (p->cpl) {
        for (int k = 0; k < HEAAC_MAX_CCE; k++) {
            if (p->cpl->d_state[k]) (void)hipFree(p->cpl->d_state[k]);
            if (p->cpl->d_pred[k]) (void)hipFree(p->cpl->d_pred[k]);
            if (p->cpl->d_ret[k]) (void)hipFree(p->cpl->d_ret[k]);
            free(p->cpl->sst[k]);
        }
        if (p->cpl->d_state_tmp) (void)hipFree(p->cpl->d_state_tmp);
        if (p->cpl->d_plane) (void)hipFree(p->cpl->d_plane);
        free(p->cpl->seen);
        free(p->cpl);
    }
    if (p->d_rng) (void)hipFree(p->d_rng);
    if (p->d_hdr) (void)hipFree(p->d_hdr);
    if (p->d_park) (void)hipFree(p->d_park);
    if (p->run) (void)hipStreamDestroy(p->run);
    heaac_sbr_table_destroy(p->tab);
    free(p->ast); free(p->lay);
    heaac_device_destroy(p->dev);
    free(p);
}

extern "C" int heaac_layout_pipeline_create(HeaacLayoutPipeline **out, const HeaacAacConfig *aac, const HeaacAacLayout *layout,
                                            size_t n, int threads)
{
    if (!out) return HEAAC_ERR_ARG;
    *out = NULL;
    if (!aac || !layout || !n || layout->n_elements < 1 || layout->n_elements > HEAAC_MAX_ELEMENTS ||
        layout->channels < 1 || layout->channels > HEAAC_MAX_PCM_PLANES || aac->sampling_index < 0 || aac->sampling_index > 12 ||
        (aac->sbr != 0 && aac->sbr != 1))                 // implicit signalling (-1) is settled per stream by its first unit
        return HEAAC_ERR_ARG;
    // coupling channel elements: slots 0 .. K-1 of the layout's list
    int n_cce_slots = 0;
    for (int id = 0; id < 16; id++)
        if (layout->slot_of[HEAAC_ELEM_CCE][id]) {
            if (layout->slot_of[HEAAC_ELEM_CCE][id] > n_cce_slots) n_cce_slots = layout->slot_of[HEAAC_ELEM_CCE][id];
        }
    if (n_cce_slots > HEAAC_MAX_CCE) return HEAAC_ERR_ARG;
    {
        int outs = 0;
        for (int k = 0; k < layout->n_elements; k++)
            outs += aac->sbr == 1 && aac->ps != 0 && layout->elem[k].type == HEAAC_ELEM_SCE ? 2 : layout->elem[k].channels;
        if (outs > HEAAC_MAX_PCM_PLANES) return HEAAC_ERR_ARG;
    }
    HeaacLayoutPipeline *p = (HeaacLayoutPipeline *)calloc(1, sizeof(*p));
    if (!p) return HEAAC_ERR_NOMEM;
    p->aac = *aac;
    p->layout = *layout;
    p->ne = layout->n_elements;
    p->n = n;
    p->he = aac->sbr == 1;
    const int mode = p->he ? heaac_sbr_output_mode(aac) : 0;
    if (mode < 0) { free(p); return HEAAC_ERR_ARG; }
    p->downsampled = mode;
    p->len = p->he && !mode ? 2048 : 1024;
    p->main_profile = aac->object_type == HEAAC_AOT_AAC_MAIN;
    int rc = heaac_device_create(&p->dev, n);
    if (rc != HEAAC_OK) { free(p); return rc; }
    bool ok = hipStreamCreateWithFlags(&p->run, hipStreamNonBlocking) == hipSuccess;
    HeaacPredictorState *ps = NULL;
    // explicit SBR with Parametric Stereo on (or left open, which decode_audio_specific_config reads as on,
    // aacdec.c:476-477): every single channel element gives two channels (che_configure :203-206; codec_layout.hip)
    const bool ps_sce = aac->sbr == 1 && aac->ps != 0;
    for (int k = 0; ok && k < p->ne; k++) {
        LpElem &e = p->e[k];
        e.channels = layout->elem[k].channels;
        e.cfg = p->he ? (e.channels == 2 ? HEAAC_CFG_HEV1 : HEAAC_CFG_HEV1_MONO)
                      : (e.channels == 2 ? HEAAC_CFG_LC_STEREO : HEAAC_CFG_LC_MONO);
        e.out = e.channels;
        if (ps_sce && layout->elem[k].type == HEAAC_ELEM_SCE) { e.cfg = HEAAC_CFG_HEV2; e.out = 2; }
        e.first_out = k ? p->e[k - 1].first_out + p->e[k - 1].out : 0;
        p->out_channels = e.first_out + e.out;
        e.words = e.cfg == HEAAC_CFG_HEV1 ? HEAAC_STATE_WORDS_HEV1 : e.cfg == HEAAC_CFG_HEV1_MONO ? HEAAC_STATE_WORDS_HEV1_MONO :
                  e.cfg == HEAAC_CFG_HEV2 ? HEAAC_STATE_WORDS_HEV2 :
                  e.cfg == HEAAC_CFG_LC_STEREO ? HEAAC_STATE_WORDS_LC_STEREO : HEAAC_STATE_WORDS_LC_MONO;
        ok = lp_devmem((void **)&e.d_state, n * (size_t)e.words * 4) && hipMemset(e.d_state, 0, n * (size_t)e.words * 4) == hipSuccess &&
             lp_devmem((void **)&e.d_f32, n * (size_t)e.out * p->len * 4);
        if (ok && p->he) {
            e.sst = (HeaacSbrStream *)malloc(n * heaac_sbr_stream_bytes());
            ok = e.sst != NULL;
            if (ok) heaac_sbr_stream_init(e.sst, n);
        }
        if (ok && p->main_profile) {
            // reset_predict_state (aacdec.c:507-515) for every predictor of every channel
            const size_t np = n * (size_t)e.channels * HEAAC_MAX_PREDICTORS;
            ps = (HeaacPredictorState *)calloc(np, sizeof(*ps));
            ok = ps != NULL && lp_devmem((void **)&e.d_pred, np * sizeof(*ps));
            if (ok) {
                for (size_t i = 0; i < np; i++) ps[i].var0 = ps[i].var1 = 1.0f;
                ok = hipMemcpy(e.d_pred, ps, np * sizeof(*ps), hipMemcpyHostToDevice) == hipSuccess;
            }
            free(ps);
            ps = NULL;
        }
    }
    if (ok && n_cce_slots) {
        const int K = n_cce_slots;
        LpCoupled *c = p->cpl = (LpCoupled *)calloc(1, sizeof(LpCoupled));
        ok = c != NULL;
        if (ok) {
            c->K = K;
            c->words = p->he ? HEAAC_STATE_WORDS_HEV1_MONO : 512;
            c->seen = (unsigned char *)calloc(n * K, 1);
            ok = c->seen != NULL && lp_devmem((void **)&c->d_state_tmp, n * (size_t)c->words * 4) &&
                 lp_devmem((void **)&c->d_plane, n * (size_t)p->len * 4);
        }
        HeaacPredictorState *reset = NULL;
        if (ok && p->main_profile) {
            reset = (HeaacPredictorState *)calloc(n * HEAAC_MAX_PREDICTORS, sizeof(*reset));
            ok = reset != NULL;
            for (size_t i = 0; ok && i < n * HEAAC_MAX_PREDICTORS; i++) reset[i].var0 = reset[i].var1 = 1.0f;
        }
        for (int k = 0; ok && k < K; k++) {
            const size_t sb = n * (size_t)c->words * 4;
            if (p->he) {
                c->sst[k] = (HeaacSbrStream *)malloc(n * heaac_sbr_stream_bytes());
                ok = c->sst[k] != NULL;
                if (ok) heaac_sbr_stream_init(c->sst[k], n);
            }
            ok = ok && lp_devmem((void **)&c->d_state[k], sb) && hipMemset(c->d_state[k], 0, sb) == hipSuccess &&
                 lp_devmem((void **)&c->d_ret[k], n * (size_t)p->len * 4) &&
                 (!reset || (lp_devmem((void **)&c->d_pred[k], n * HEAAC_MAX_PREDICTORS * sizeof(*reset)) &&
                             hipMemcpy(c->d_pred[k], reset, n * HEAAC_MAX_PREDICTORS * sizeof(*reset), hipMemcpyHostToDevice) == hipSuccess));
        }
        free(reset);
        for (int q = 0; q < LP_DEPTH && ok; q++) {
            LpSet *s = &p->set[q];
            for (int e = 0; e < p->ne && ok; e++)
                ok = lp_pinned((void **)&s->h_cce[e], n * K * sizeof(HeaacCceFrame)) && lp_devmem((void **)&s->d_cce[e], n * K * sizeof(HeaacCceFrame));
            for (int k = 0; k < K && ok; k++)
                ok = lp_pinned((void **)&s->h_ccoef[k], n * 4096) && lp_devmem((void **)&s->d_ccoef[k], n * 4096) &&
                     lp_pinned((void **)&s->h_cics[k], n * sizeof(HeaacIcs)) && lp_devmem((void **)&s->d_cics[k], n * sizeof(HeaacIcs)) &&
                     lp_pinned((void **)&s->h_ctools[k], n * sizeof(HeaacToolsFrame)) && lp_devmem((void **)&s->d_ctools[k], n * sizeof(HeaacToolsFrame)) &&
                     (!p->he || (lp_pinned((void **)&s->h_csbr[k], n * sizeof(HeaacSbrFrame)) && lp_devmem((void **)&s->d_csbr[k], n * sizeof(HeaacSbrFrame))));
            const size_t ng = (size_t)p->ne * K * HEAAC_MAX_CCE_LINKS * n * (p->len / 1024);
            ok = ok && lp_devmem((void **)&s->d_ccoef_fm, n * K * 4096) &&
                 lp_pinned((void **)&s->h_gain, ng * sizeof(HeaacCoupling)) && lp_devmem((void **)&s->d_gain, ng * sizeof(HeaacCoupling)) &&
                 (s->cpos = (unsigned char *)calloc(n * K, 3)) != NULL;
        }
    }
    for (int q = 0; q < LP_DEPTH && ok; q++) {
        LpSet *s = &p->set[q];
        for (int k = 0; k < p->ne && ok; k++) {
            const size_t nc = n * (size_t)p->e[k].channels;
            ok = lp_pinned((void **)&s->h_coeffs[k], nc * 4096) && lp_devmem((void **)&s->d_coeffs[k], nc * 4096) &&
                 lp_pinned((void **)&s->h_ics[k], nc * sizeof(HeaacIcs)) && lp_devmem((void **)&s->d_ics[k], nc * sizeof(HeaacIcs)) &&
                 lp_pinned((void **)&s->h_tools[k], n * sizeof(HeaacToolsFrame)) && lp_devmem((void **)&s->d_tools[k], n * sizeof(HeaacToolsFrame)) &&
                 (!p->he || (lp_pinned((void **)&s->h_sbr[k], n * sizeof(HeaacSbrFrame)) && lp_devmem((void **)&s->d_sbr[k], n * sizeof(HeaacSbrFrame)))) &&
                 (p->e[k].cfg != HEAAC_CFG_HEV2 ||
                  (lp_pinned((void **)&s->h_ps[k], n * sizeof(HeaacPsFrame)) && lp_devmem((void **)&s->d_ps[k], n * sizeof(HeaacPsFrame))));
        }
        const size_t pcm_bytes = n * (size_t)p->out_channels * p->len * 2;
        ok = ok && lp_pinned((void **)&s->h_pcm, pcm_bytes) && lp_devmem((void **)&s->d_pcm, pcm_bytes) &&
             (s->failed = (unsigned char *)calloc(n, 1)) != NULL && (s->seq = (unsigned char *)calloc(n * p->ne, 1)) != NULL &&
             lp_pinned((void **)&s->h_list, (4 + 2 * (size_t)n_cce_slots) * n * sizeof(unsigned)) &&
             lp_devmem((void **)&s->d_list, (4 + 2 * (size_t)n_cce_slots) * n * sizeof(unsigned)) &&
             hipEventCreate(&s->done) == hipSuccess;
    }
    ok = ok && lp_devmem((void **)&p->d_rng, n * 4) && lp_devmem((void **)&p->d_hdr, LP_MAX_HDRS * sizeof(HeaacSbrHeader));
    if (ok) {
        int32_t *seed = (int32_t *)malloc(n * 4);
        ok = seed != NULL;
        if (ok) {
            for (size_t i = 0; i < n; i++) seed[i] = 0x1f2e3d4c;       // ac->random_state, aacdec.c:558
            ok = hipMemcpy(p->d_rng, seed, n * 4, hipMemcpyHostToDevice) == hipSuccess;
            free(seed);
        }
    }
    p->tab = heaac_sbr_table_create(LP_MAX_HDRS);
    p->ast = (HeaacAacStream *)calloc(n * p->ne, sizeof(HeaacAacStream));
    p->lay = (HeaacAacLayout *)malloc(n * sizeof(HeaacAacLayout));
    ok = ok && p->tab && p->ast && p->lay;
    if (ok) {
        for (size_t i = 0; i < n; i++) p->lay[i] = *layout;
        // the null header (table entry 0) is what frames before their element's first header point at
        ok = hipMemcpy(p->d_hdr, heaac_sbr_table_data(p->tab), sizeof(HeaacSbrHeader), hipMemcpyHostToDevice) == hipSuccess;
        p->hdr_uploaded = 1;
    }
    if (ok) {
        if (threads <= 0) {
            long online = sysconf(_SC_NPROCESSORS_ONLN);
            threads = online < 1 ? 1 : online > 32 ? 32 : (int)online;
        }
        if (threads > 256) threads = 256;
        if ((size_t)threads > n) threads = (int)n;
        p->threads = threads;
        pthread_mutex_init(&p->mu, NULL);
        pthread_cond_init(&p->cv_go, NULL);
        pthread_cond_init(&p->cv_done, NULL);
        p->tid = (pthread_t *)calloc(threads, sizeof(pthread_t));
        ok = p->tid != NULL;
        for (int t = 1; t < threads && ok; t++) {          // slice 0 is parsed by the submitting thread
            LpWorkerArg *a = (LpWorkerArg *)malloc(sizeof(*a));
            if (!a) { ok = false; break; }
            a->p = p; a->w = t;
            if (pthread_create(&p->tid[t], NULL, lp_worker, a) != 0) { free(a); p->tid[t] = 0; p->threads = t; break; }
        }
    }
    if (!ok) { heaac_layout_pipeline_destroy(p); return HEAAC_ERR_NOMEM; }
    *out = p;
    return HEAAC_OK;
}

#define LP_HIP(x) do { if ((x) != hipSuccess) return HEAAC_ERR_HIP; } while (0)

// Rows of the failed streams to the parking area and back (or zeroed): one block per listed (stream, parking row)
// pair -- a handful of launches per tick however many units are damaged (pipeline.hip k_rows).
//   mode 0: park[row] = rows[stream];  1: rows[stream] = park[row];  2: rows[stream] = 0
__global__ void k_lp_rows(const unsigned *__restrict__ list, unsigned *rows, unsigned *park, unsigned long long row_words, int mode)
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

// rows of every failed stream: element states, predictors, noise generator -- to / from the parking area, which holds
// one region of `cap` rows per array.  n_all pairs at d_list: every failed stream; n_full at d_list + 2 n: those whose
// generator and predictors stay put as well (failed == 1; 2: the tools' side of the stream moves on).
static int lp_park(HeaacLayoutPipeline *p, LpSet *s, unsigned n_all, unsigned n_full, size_t cap, int restore)
{
    char *region = (char *)p->d_park;
    for (int k = 0; k < p->ne; k++) {
        const LpElem &e = p->e[k];
        const size_t sb = (size_t)e.words * 4, pb = (size_t)e.channels * HEAAC_MAX_PREDICTORS * sizeof(HeaacPredictorState);
        hipLaunchKernelGGL(k_lp_rows, dim3(n_all), dim3(256), 0, p->run, s->d_list, (unsigned *)e.d_state, (unsigned *)region,
                           (unsigned long long)e.words, restore);
        region += cap * sb;
        if (e.d_pred) {
            if (n_full)
                hipLaunchKernelGGL(k_lp_rows, dim3(n_full), dim3(256), 0, p->run, s->d_list + 2 * p->n, (unsigned *)e.d_pred, (unsigned *)region,
                                   (unsigned long long)(pb / 4), restore);
            region += cap * pb;
        }
    }
    if (n_full)
        hipLaunchKernelGGL(k_lp_rows, dim3(n_full), dim3(64), 0, p->run, s->d_list + 2 * p->n, (unsigned *)p->d_rng, (unsigned *)region, 1ull, restore);
    return hipGetLastError() == hipSuccess ? HEAAC_OK : HEAAC_ERR_HIP;
}

extern "C" int heaac_layout_pipeline_submit(HeaacLayoutPipeline *p, const uint8_t *const *au, const int *size, int *status)
{
    if (!p || !au || !size) return HEAAC_ERR_ARG;
    if (p->submitted - p->collected >= LP_DEPTH) return HEAAC_ERR_ARG;
    LpSet *s = &p->set[p->submitted % LP_DEPTH];
    // (the set's buffers are free: its last tick has been collected, which waited for its `done`)
    pthread_mutex_lock(&p->mu);
    p->job_au = au; p->job_size = size; p->job_status = status; p->job_set = s;
    p->pending = p->threads - 1;
    p->generation++;
    pthread_cond_broadcast(&p->cv_go);
    pthread_mutex_unlock(&p->mu);
    lp_parse_slice(p, 0);
    pthread_mutex_lock(&p->mu);
    while (p->pending > 0) pthread_cond_wait(&p->cv_done, &p->mu);
    pthread_mutex_unlock(&p->mu);

    const size_t n = p->n;
    const int ne = p->ne;
    // the element order of the pipeline's streams: the first good unit sets it, a stream that deviates is dropped for the tick
    size_t n_failed = 0;
    for (size_t i = 0; i < n; i++) {
        if (s->failed[i] == 2) {
            // the elements a refused unit got through must stand where the pipeline's order has them: the generator
            // runs through the elements in that order
            bool same = p->have_order != 0;
            for (int k = 0; same && k < ne; k++)
                same = s->seq[i * ne + k] == 0xff || (s->seq[i * ne + k] < ne && p->order[s->seq[i * ne + k]] == k);
            if (!same) { s->failed[i] = 1; neutral(p, s, i); }
        }
        if (s->failed[i]) { n_failed++; continue; }
        if (!p->have_order) {
            for (int k = 0; k < ne; k++) p->order[s->seq[i * ne + k] < ne ? s->seq[i * ne + k] : 0] = (unsigned char)k;
            p->have_order = 1;
        }
        bool same = true;
        for (int k = 0; k < ne; k++) same = same && s->seq[i * ne + k] < ne && p->order[s->seq[i * ne + k]] == k;
        if (!same) {
            s->failed[i] = 1;
            n_failed++;
            if (status) status[i] = HEAAC_PARSE_ERR_UNSUPPORTED;
            neutral(p, s, i);
        }
    }
    // ... and where the coupling elements stand among them, tick by tick: the first good stream of the tick says, the
    // others must agree (the coupling POINT may differ from stream to stream)
    const int K = p->cpl ? p->cpl->K : 0;
    unsigned char cpat[HEAAC_MAX_CCE][3];
    int have_cpat = 0, n_cce_tick = 0;
    memset(cpat, 0, sizeof(cpat));
    for (size_t i = 0; K && i < n; i++) {
        if (s->failed[i]) continue;
        const unsigned char *cp = s->cpos + i * K * 3;
        if (!have_cpat) { memcpy(cpat, cp, (size_t)K * 3); have_cpat = 1; continue; }
        bool same = true;
        for (int k = 0; k < K; k++)
            same = same && cp[3 * k] == cpat[k][0] && (!cp[3 * k] || (cp[3 * k + 1] == cpat[k][1] && cp[3 * k + 2] == cpat[k][2]));
        if (!same) {
            s->failed[i] = 1;
            n_failed++;
            if (status) status[i] = HEAAC_PARSE_ERR_UNSUPPORTED;
            neutral(p, s, i);
        }
    }
    for (int k = 0; k < K; k++) n_cce_tick += cpat[k][0];
    const size_t have = heaac_sbr_table_count(p->tab);
    if (have > LP_MAX_HDRS) return HEAAC_ERR_ARG;
    // H2D (the run stream carries everything: the tick before has the GPU meanwhile)
    if (have > p->hdr_uploaded) {
        LP_HIP(hipMemcpyAsync(p->d_hdr + p->hdr_uploaded, heaac_sbr_table_data(p->tab) + p->hdr_uploaded,
                              (have - p->hdr_uploaded) * sizeof(HeaacSbrHeader), hipMemcpyHostToDevice, p->run));
        p->hdr_uploaded = have;
    }
    for (int k = 0; k < ne; k++) {
        const size_t nc = n * (size_t)p->e[k].channels;
        LP_HIP(hipMemcpyAsync(s->d_coeffs[k], s->h_coeffs[k], nc * 4096, hipMemcpyHostToDevice, p->run));
        LP_HIP(hipMemcpyAsync(s->d_ics[k], s->h_ics[k], nc * sizeof(HeaacIcs), hipMemcpyHostToDevice, p->run));
        LP_HIP(hipMemcpyAsync(s->d_tools[k], s->h_tools[k], n * sizeof(HeaacToolsFrame), hipMemcpyHostToDevice, p->run));
        if (p->he) LP_HIP(hipMemcpyAsync(s->d_sbr[k], s->h_sbr[k], n * sizeof(HeaacSbrFrame), hipMemcpyHostToDevice, p->run));
        if (s->d_ps[k]) LP_HIP(hipMemcpyAsync(s->d_ps[k], s->h_ps[k], n * sizeof(HeaacPsFrame), hipMemcpyHostToDevice, p->run));
    }
    if (n_cce_tick) {
        for (int e = 0; e < ne; e++)
            LP_HIP(hipMemcpyAsync(s->d_cce[e], s->h_cce[e], n * K * sizeof(HeaacCceFrame), hipMemcpyHostToDevice, p->run));
        for (int k = 0; k < K; k++) {
            LP_HIP(hipMemcpyAsync(s->d_ccoef[k], s->h_ccoef[k], n * 4096, hipMemcpyHostToDevice, p->run));
            LP_HIP(hipMemcpyAsync(s->d_cics[k], s->h_cics[k], n * sizeof(HeaacIcs), hipMemcpyHostToDevice, p->run));
            LP_HIP(hipMemcpyAsync(s->d_ctools[k], s->h_ctools[k], n * sizeof(HeaacToolsFrame), hipMemcpyHostToDevice, p->run));
            if (p->he) LP_HIP(hipMemcpyAsync(s->d_csbr[k], s->h_csbr[k], n * sizeof(HeaacSbrFrame), hipMemcpyHostToDevice, p->run));
        }
    }
    unsigned n_all = 0, n_full = 0;
    if (n_failed) {
        size_t row = 4;
        for (int k = 0; k < ne; k++)
            row += (size_t)p->e[k].words * 4 + (p->e[k].d_pred ? (size_t)p->e[k].channels * HEAAC_MAX_PREDICTORS * sizeof(HeaacPredictorState) : 0);
        if (n_failed > p->park_rows) {
            LP_HIP(hipStreamSynchronize(p->run));
            if (p->d_park) (void)hipFree(p->d_park);
            p->d_park = NULL; p->park_bytes = 0; p->park_rows = 0;
            size_t rows = 16;
            while (rows < n_failed) rows *= 2;
            if (rows > n) rows = n;
            if (!lp_devmem((void **)&p->d_park, rows * row)) return HEAAC_ERR_NOMEM;
            p->park_bytes = rows * row;
            p->park_rows = rows;
        }
        unsigned *list_all = s->h_list, *list_full = s->h_list + 2 * n;
        for (size_t i = 0; i < n; i++) {
            if (!s->failed[i]) continue;
            if (s->failed[i] == 1) { list_full[2 * n_full] = (unsigned)i; list_full[2 * n_full + 1] = n_all; n_full++; }
            list_all[2 * n_all] = (unsigned)i; list_all[2 * n_all + 1] = n_all; n_all++;
        }
        LP_HIP(hipMemcpyAsync(s->d_list, list_all, 2 * n_all * sizeof(unsigned), hipMemcpyHostToDevice, p->run));
        if (n_full) LP_HIP(hipMemcpyAsync(s->d_list + 2 * n, list_full, 2 * n_full * sizeof(unsigned), hipMemcpyHostToDevice, p->run));
        const int rc = lp_park(p, s, n_all, n_full, p->park_rows, 0);
        if (rc != HEAAC_OK) return rc;
    }
    // the spectral tools of the elements in bitstream order (one noise generator per stream).  A coupling element's
    // tools as a whole at its place in the stream (nothing couples INTO it); with coupling elements in the tick an
    // output element's first half there and its second half -- coupling, TNS, coupling -- once every coupling
    // element is through (codec_layout.hip; spectral_to_sample walks the element types downwards, aacdec.c:1907).
    for (int q = 0; q <= ne; q++) {
        for (int seq = 0; seq < n_cce_tick; seq++)
            for (int k = 0; k < K; k++) {
                if (!cpat[k][0] || cpat[k][1] != q || cpat[k][2] != seq) continue;
                const int rc = heaac_spectral_tools_batch_ex(p->dev, 1, HEAAC_TOOLS_ALL, s->d_ccoef[k], s->d_ctools[k], p->d_rng, p->d_rng,
                                                             p->cpl->d_pred[k], p->cpl->d_pred[k], NULL, NULL, 0, n, (void *)p->run);
                if (rc != HEAAC_OK) return rc;
            }
        if (q == ne) break;
        const int k = p->have_order ? p->order[q] : q;
        const LpElem &e = p->e[k];
        const int rc = heaac_spectral_tools_batch_ex(p->dev, e.channels, n_cce_tick ? HEAAC_TOOLS_PRE : HEAAC_TOOLS_ALL, s->d_coeffs[k],
                                                     s->d_tools[k], p->d_rng, p->d_rng, e.d_pred, e.d_pred, NULL, NULL, 0, n, (void *)p->run);
        if (rc != HEAAC_OK) return rc;
    }
    if (n_cce_tick) {
        // the coupling channels' spectra frame-major, K slots per stream, as the second half reads them
        for (int k = 0; k < K; k++)
            LP_HIP(hipMemcpy2DAsync(s->d_ccoef_fm + (size_t)k * 1024, (size_t)K * 4096, s->d_ccoef[k], 4096, 4096, n,
                                    hipMemcpyDeviceToDevice, p->run));
        for (int k = 0; k < ne; k++) {
            const LpElem &e = p->e[k];
            const int rc = heaac_spectral_tools_batch_ex(p->dev, e.channels, HEAAC_TOOLS_POST, s->d_coeffs[k], s->d_tools[k], NULL, NULL,
                                                         NULL, NULL, s->d_cce[k], s->d_ccoef_fm, K, n, (void *)p->run);
            if (rc != HEAAC_OK) return rc;
        }
        // the coupling channels that couple behind the IMDCT: their own IMDCT first (type 2 before types 1 and 0).  Only
        // the streams whose element couples there this tick may move its overlap state: where all do the call works in
        // place, where some do the others' rows are taken from a scratch copy of the state.
        for (int k = 0; k < K; k++) {
            if (!cpat[k][0]) continue;
            size_t after = 0, live = 0;
            for (size_t i = 0; i < n; i++) {
                live += !s->failed[i];
                after += !s->failed[i] && s->h_cce[0][i * K + k].present && s->h_cce[0][i * K + k].coupling_point == HEAAC_CC_AFTER_IMDCT;
            }
            if (!after) continue;
            float *st = p->cpl->d_state[k];
            const size_t words = (size_t)p->cpl->words;
            const bool all = after == n && live == n;
            const int rc = p->he
                ? heaac_he_decode_batch_ex(p->dev, HEAAC_CFG_HEV1_MONO, p->downsampled ? HEAAC_HE_DOWNSAMPLED : 0, s->d_ccoef[k], s->d_cics[k],
                                           s->d_csbr[k], p->d_hdr, LP_MAX_HDRS, NULL, st, all ? st : p->cpl->d_state_tmp, p->cpl->d_ret[k],
                                           HEAAC_PCM_F32_PLANAR, n, (void *)p->run)
                : heaac_lc_decode_batch(p->dev, 1, s->d_ccoef[k], s->d_cics[k], st, all ? st : p->cpl->d_state_tmp,
                                        p->cpl->d_ret[k], HEAAC_PCM_F32_PLANAR, n, (void *)p->run);
            if (rc != HEAAC_OK) return rc;
            if (!all) {
                // rows of the streams that couple there, from the scratch copy (row i of it is stream i's)
                unsigned *lst = s->h_list + (4 + 2 * (size_t)k) * n, cnt = 0;
                for (size_t i = 0; i < n; i++)
                    if (!s->failed[i] && s->h_cce[0][i * K + k].present && s->h_cce[0][i * K + k].coupling_point == HEAAC_CC_AFTER_IMDCT) {
                        lst[2 * cnt] = lst[2 * cnt + 1] = (unsigned)i;
                        cnt++;
                    }
                unsigned *dl = s->d_list + (4 + 2 * (size_t)k) * n;
                LP_HIP(hipMemcpyAsync(dl, lst, 2 * cnt * sizeof(unsigned), hipMemcpyHostToDevice, p->run));
                hipLaunchKernelGGL(k_lp_rows, dim3(cnt), dim3(256), 0, p->run, dl, (unsigned *)st, (unsigned *)p->cpl->d_state_tmp,
                                   (unsigned long long)words, 1);
                LP_HIP(hipGetLastError());
            }
        }
    }
    HeaacPlaneRef planes[HEAAC_MAX_PCM_PLANES];
    for (int k = 0; k < ne; k++) {
        const LpElem &e = p->e[k];
        const int rc = p->he
            ? heaac_he_decode_batch_ex(p->dev, e.cfg, p->downsampled ? HEAAC_HE_DOWNSAMPLED : 0, s->d_coeffs[k], s->d_ics[k], s->d_sbr[k],
                                       p->d_hdr, LP_MAX_HDRS, s->d_ps[k], e.d_state, e.d_state, e.d_f32, HEAAC_PCM_F32_PLANAR, n, (void *)p->run)
            : heaac_lc_decode_batch(p->dev, e.channels, s->d_coeffs[k], s->d_ics[k], e.d_state, e.d_state, e.d_f32,
                                    HEAAC_PCM_F32_PLANAR, n, (void *)p->run);
        if (rc != HEAAC_OK) return rc;
        // every AFTER_IMDCT element in tag order, every gain list that lands on this element (apply_channel_coupling
        // :1870-1898; apply_independent_coupling :1849-1862): one batched call per list, gains per stream
        for (int kc = 0; kc < K && n_cce_tick; kc++) {
            if (!cpat[kc][0]) continue;
            // The batched op adds a [frames][1024] coupling signal into [frames][channels][1024] targets.  Planes of 1024
            // samples are that as they stand; planes of 2048 (behind SBR) are two such frames per stream with the stream's
            // gain twice -- for an element of one plane directly, for one of two planes channel by channel on a gathered
            // copy of the plane (apply_independent_coupling runs over 1024 << sbr samples, aacdec.c:1849-1862).
            const int sub = p->len / 1024;
            const bool direct = sub == 1 || e.out == 1;
            for (int l = 0; l < HEAAC_MAX_CCE_LINKS; l++) {
                HeaacCoupling *g = s->h_gain + (((size_t)k * K + kc) * HEAAC_MAX_CCE_LINKS + l) * n * sub;
                size_t used[2] = { 0, 0 };
                for (size_t i = 0; i < n; i++) {
                    const HeaacCceFrame &r = s->h_cce[k][i * K + kc];
                    memset(&g[i * sub], 0, sub * sizeof(g[0]));
                    if (s->failed[i] || !r.present || r.coupling_point != HEAAC_CC_AFTER_IMDCT || l >= r.n_links) continue;
                    const int tch = r.link[l].target_ch < e.channels ? r.link[l].target_ch : 0;
                    for (int q = 0; q < sub; q++) {
                        // (gathered planes are one-channel frames: the gain sits in channel 0 of the record)
                        g[i * sub + q].on[direct ? tch : 0] = 1;
                        g[i * sub + q].gain[direct ? tch : 0] = r.link[l].gain[0];
                    }
                    used[tch]++;
                }
                if (!used[0] && !used[1]) continue;
                HeaacCoupling *dg = s->d_gain + (g - s->h_gain);
                if (direct) {
                    LP_HIP(hipMemcpyAsync(dg, g, n * sub * sizeof(HeaacCoupling), hipMemcpyHostToDevice, p->run));
                    const int rc2 = heaac_couple_after_imdct_batch(p->dev, sub == 1 ? e.out : 1, e.d_f32, p->cpl->d_ret[kc], dg, NULL,
                                                                   n * sub, (void *)p->run);
                    if (rc2 != HEAAC_OK) return rc2;
                    continue;
                }
                // two planes of 2048: a gain list lands on ONE channel per stream, and which one may differ from stream to
                // stream -- one pass per channel, each with the gains of the streams that target it
                for (int tc2 = 0; tc2 < 2; tc2++) {
                    if (!used[tc2]) continue;
                    if (used[tc2 ^ 1]) {
                        // mixed targets: this pass takes only the streams whose list lands on tc2
                        for (size_t i = 0; i < n; i++) {
                            const HeaacCceFrame &r = s->h_cce[k][i * K + kc];
                            const bool mine = !s->failed[i] && r.present && r.coupling_point == HEAAC_CC_AFTER_IMDCT && l < r.n_links &&
                                              (r.link[l].target_ch < e.channels ? r.link[l].target_ch : 0) == tc2;
                            for (int q = 0; q < sub; q++) {
                                g[i * sub + q].on[0] = mine;
                                g[i * sub + q].gain[0] = mine ? r.link[l].gain[0] : 0.0f;
                            }
                        }
                        LP_HIP(hipStreamSynchronize(p->run));      // (the staging area is about to be rewritten for the other channel)
                    }
                    LP_HIP(hipMemcpyAsync(dg, g, n * sub * sizeof(HeaacCoupling), hipMemcpyHostToDevice, p->run));
                    const size_t row = (size_t)p->len * 4;
                    LP_HIP(hipMemcpy2DAsync(p->cpl->d_plane, row, e.d_f32 + (size_t)tc2 * p->len, 2 * row, row, n, hipMemcpyDeviceToDevice, p->run));
                    const int rc2 = heaac_couple_after_imdct_batch(p->dev, 1, p->cpl->d_plane, p->cpl->d_ret[kc], dg, NULL, n * sub, (void *)p->run);
                    if (rc2 != HEAAC_OK) return rc2;
                    LP_HIP(hipMemcpy2DAsync(e.d_f32 + (size_t)tc2 * p->len, 2 * row, p->cpl->d_plane, row, row, n, hipMemcpyDeviceToDevice, p->run));
                    if (used[tc2 ^ 1]) LP_HIP(hipStreamSynchronize(p->run));
                }
            }
        }
        for (int c = 0; c < e.out; c++) {
            planes[e.first_out + c].d_base = e.d_f32 + (size_t)c * p->len;
            planes[e.first_out + c].frame_stride = (size_t)e.out * p->len;
        }
    }
    int rc = heaac_pcm_interleave_batch(p->dev, p->out_channels, planes, p->len, HEAAC_PCM_S16_INTERLEAVED, s->d_pcm, n, (void *)p->run);
    if (rc != HEAAC_OK) return rc;
    const size_t pcm_row = (size_t)p->out_channels * p->len;
    if (n_failed) {
        rc = lp_park(p, s, n_all, n_full, p->park_rows, 1);
        if (rc != HEAAC_OK) return rc;
        hipLaunchKernelGGL(k_lp_rows, dim3(n_all), dim3(256), 0, p->run, s->d_list, (unsigned *)s->d_pcm, (unsigned *)nullptr,
                           (unsigned long long)(pcm_row / 2), 2);      // (len is a multiple of 1024: whole 32-bit words)
        LP_HIP(hipGetLastError());
    }
    LP_HIP(hipMemcpyAsync(s->h_pcm, s->d_pcm, n * pcm_row * 2, hipMemcpyDeviceToHost, p->run));
    LP_HIP(hipEventRecord(s->done, p->run));
    s->used = 1;
    p->submitted++;
    return HEAAC_OK;
}

extern "C" int heaac_layout_pipeline_channels(const HeaacLayoutPipeline *p) { return p ? p->out_channels : 0; }

extern "C" int heaac_layout_pipeline_collect(HeaacLayoutPipeline *p, const int16_t **pcm)
{
    if (!p || !pcm || p->collected == p->submitted) return HEAAC_ERR_ARG;
    LpSet *s = &p->set[p->collected % LP_DEPTH];
    LP_HIP(hipEventSynchronize(s->done));
    *pcm = s->h_pcm;
    p->collected++;
    return HEAAC_OK;
}
