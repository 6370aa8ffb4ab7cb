// codec_layout.hip -- aac_decode_frame for multi-element layouts behind the AVCodec surface (shim.hip):
// the element loop is heaac_aac_parse_frame_layout (aac_parse.c), spectral_to_sample (aacdec.c:1903-1933) is one
// decode call per element on that element's own state record, float_to_int16_interleave over output_data[]
// (:2096-2097) is heaac_pcm_interleave_batch over the elements' float planes in layout order.
//   * one noise generator for the stream, run through the elements in bitstream order (decode_spectrum_and_dequant
//     draws from ac->random_state as it parses, :1049-1054);
//   * SBR per element (che->sbr): an element's payload is the fill element directly behind it; once the stream has
//     SBR (explicitly, or implicitly by a payload in the FIRST access unit, :1666-1675) every element goes through
//     ff_sbr_apply, with a start = 0 record ("pure upsampling") where it has no payload; a payload behind an LFE, or
//     with another fill / data stream element between it and its element, is read for its header and switches the
//     element's SBR off (aacsbr.c:996-1000);
//   * an access unit that leaves an element of the layout out is refused: the reference transforms whatever that
//     element's buffers still hold from an earlier frame, which no record of this path carries;
//   * coupling channel elements: those the program config element names
//     (che_configure allocates no others).  They are individual channel streams of their own -- tools, and an
//     IMDCT when they couple AFTER_IMDCT -- processed before their targets (spectral_to_sample walks the element
//     types downwards, :1907); dependent coupling sits around a target's TNS, independent coupling behind its
//     IMDCT (:1911-1930), coupling elements in ascending tag order (apply_channel_coupling :1876).  One that an
//     earlier access unit carried and this one leaves out is refused for the same reason as an output element.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include "codec_layout.h"

#define LAY_MAX_HDRS 64
#define CCE_STATE_WORDS HEAAC_STATE_WORDS_HEV1_MONO
#define LAY_STATE_WORDS (HEAAC_STATE_WORDS_HEV2 > HEAAC_STATE_WORDS_HEV1 ? HEAAC_STATE_WORDS_HEV2 : HEAAC_STATE_WORDS_HEV1)

struct LayElem {
    int cfg_lc, cfg_he, channels;
    int out_channels, first_out;  // planes the element gives and where they stand among the stream's; 2 for an SCE
                                  // that carries Parametric Stereo (che_configure, aacdec.c:203-206)
    HeaacPsFrame *d_ps;           // ... and its record
    HeaacAacStream ast;
    HeaacSbrStream sst;
    float *d_coeffs;              // [2][1024]
    HeaacIcs *d_ics;              // [2]
    HeaacToolsFrame *d_tools;
    float *d_state;               // LAY_STATE_WORDS (the largest of the configurations)
    HeaacPredictorState *d_pred;  // [2][672]
    HeaacSbrFrame *d_sbr;
    float *d_f32;                 // [2][2048]
};

// Host and device side of the coupling elements of one access unit
struct LayCoupled {
    HeaacCceFrame h_cce[HEAAC_MAX_ELEMENTS][HEAAC_MAX_CCE];          // per output slot: the gain lists that land there
    float h_coeffs[HEAAC_MAX_CCE][1024];
    HeaacIcs h_ics[HEAAC_MAX_CCE];
    HeaacToolsFrame h_tools[HEAAC_MAX_CCE];
    HeaacAacElementInfo h_elem[HEAAC_MAX_CCE];                       // where each stands, the SBR payload behind it
    HeaacSbrStream sst[HEAAC_MAX_CCE];                               // a coupling channel's own SBR (che->sbr)
    HeaacSbrFrame *d_sbr;         // [MAX_CCE]
    HeaacCceFrame *d_cce;         // [n_elements][MAX_CCE]
    float *d_coeffs;              // [MAX_CCE][1024]
    HeaacIcs *d_ics;
    HeaacToolsFrame *d_tools;
    float *d_state;               // [MAX_CCE][CCE_STATE_WORDS] state of the coupling channels that couple AFTER_IMDCT:
                                  // the overlap, and behind SBR everything a mono element has
    HeaacPredictorState *d_pred;  // [MAX_CCE][672]
    float *d_ret;                 // [MAX_CCE][2048] the coupling channels' own output
    HeaacCoupling *d_gain;        // [2]
    int seen[HEAAC_MAX_CCE];      // an earlier access unit carried this coupling element
};

struct HeaacLayoutDec {
    HeaacDevice *dev;
    HeaacAacConfig m4ac;
    HeaacAacLayout layout;
    int out_channels;             // avctx->channels: the layout's, plus one per SCE with Parametric Stereo
    int locked;                   // the first access unit has settled implicit SBR (output_configured == OC_LOCKED)
    int n;
    LayElem e[HEAAC_MAX_ELEMENTS];
    HeaacSbrHeaderTable *tab;
    HeaacSbrHeader hdr[LAY_MAX_HDRS];
    HeaacSbrHeader *d_hdr;
    size_t hdr_uploaded;
    int32_t *d_rng;
    int16_t *d_pcm;
    // host side of one access unit
    float *h_coeffs;              // [n][2][1024]
    HeaacIcs h_ics[HEAAC_MAX_ELEMENTS][2];
    HeaacToolsFrame *h_tools;     // [n]
    HeaacAacElementInfo h_elem[HEAAC_MAX_ELEMENTS];
    LayCoupled *cpl;              // layouts whose program config element names coupling elements
};

static void coupled_free(LayCoupled *c)
{
    if (!c) return;
    if (c->d_cce) (void)hipFree(c->d_cce);
    if (c->d_coeffs) (void)hipFree(c->d_coeffs);
    if (c->d_ics) (void)hipFree(c->d_ics);
    if (c->d_tools) (void)hipFree(c->d_tools);
    if (c->d_state) (void)hipFree(c->d_state);
    if (c->d_pred) (void)hipFree(c->d_pred);
    if (c->d_ret) (void)hipFree(c->d_ret);
    if (c->d_gain) (void)hipFree(c->d_gain);
    if (c->d_sbr) (void)hipFree(c->d_sbr);
    free(c);
}

static LayCoupled *coupled_alloc(const HeaacPredictorState *ps_reset)
{
    LayCoupled *c = (LayCoupled *)calloc(1, sizeof(*c));
    if (!c) return NULL;
    bool ok =
        hipMalloc((void **)&c->d_cce, sizeof(c->h_cce)) == hipSuccess &&
        hipMalloc((void **)&c->d_coeffs, sizeof(c->h_coeffs)) == hipSuccess &&
        hipMalloc((void **)&c->d_ics, sizeof(c->h_ics)) == hipSuccess &&
        hipMalloc((void **)&c->d_tools, sizeof(c->h_tools)) == hipSuccess &&
        hipMalloc((void **)&c->d_state, HEAAC_MAX_CCE * CCE_STATE_WORDS * 4) == hipSuccess &&
        hipMalloc((void **)&c->d_pred, HEAAC_MAX_CCE * HEAAC_MAX_PREDICTORS * sizeof(*ps_reset)) == hipSuccess &&
        hipMalloc((void **)&c->d_ret, HEAAC_MAX_CCE * 2048 * 4) == hipSuccess &&
        hipMalloc((void **)&c->d_gain, 2 * sizeof(HeaacCoupling)) == hipSuccess &&
        hipMalloc((void **)&c->d_sbr, HEAAC_MAX_CCE * sizeof(HeaacSbrFrame)) == hipSuccess &&
        hipMemset(c->d_state, 0, HEAAC_MAX_CCE * CCE_STATE_WORDS * 4) == hipSuccess;
    heaac_sbr_stream_init(c->sst, HEAAC_MAX_CCE);
    for (int k = 0; ok && k < HEAAC_MAX_CCE; k++)
        ok = hipMemcpy(c->d_pred + k * HEAAC_MAX_PREDICTORS, ps_reset, HEAAC_MAX_PREDICTORS * sizeof(*ps_reset),
                       hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { coupled_free(c); return NULL; }
    return c;
}

void heaac_layout_dec_destroy(HeaacLayoutDec *d)
{
    if (!d) return;
    for (int i = 0; i < d->n; i++) {
        LayElem &e = d->e[i];
        if (e.d_coeffs) (void)hipFree(e.d_coeffs);
        if (e.d_ics) (void)hipFree(e.d_ics);
        if (e.d_tools) (void)hipFree(e.d_tools);
        if (e.d_state) (void)hipFree(e.d_state);
        if (e.d_pred) (void)hipFree(e.d_pred);
        if (e.d_sbr) (void)hipFree(e.d_sbr);
        if (e.d_f32) (void)hipFree(e.d_f32);
        if (e.d_ps) (void)hipFree(e.d_ps);
    }
    if (d->d_hdr) (void)hipFree(d->d_hdr);
    if (d->d_rng) (void)hipFree(d->d_rng);
    if (d->d_pcm) (void)hipFree(d->d_pcm);
    coupled_free(d->cpl);
    heaac_sbr_table_destroy(d->tab);
    free(d->h_coeffs);
    free(d->h_tools);
    free(d);
}

int heaac_layout_dec_channels(const HeaacLayoutDec *d) { return d ? d->out_channels : 0; }

HeaacLayoutDec *heaac_layout_dec_create(HeaacDevice *dev, const HeaacAacConfig *m4ac, const HeaacAacLayout *layout)
{
    if (!dev || !m4ac || !layout || layout->n_elements < 1 || layout->n_elements > HEAAC_MAX_ELEMENTS ||
        layout->channels < 1 || layout->channels > HEAAC_MAX_PCM_PLANES)
        return NULL;
    // Explicitly signalled SBR with the Parametric Stereo question left open (ps = -1: only a program-config layout
    // leaves it open, mpeg4audio.c:137-139) is ps = 1 to decode_audio_specific_config (aacdec.c:476-477), and
    // che_configure then gives EVERY single channel element of the layout a second output channel (:203-206):
    // ff_sbr_apply runs ff_ps_apply on it once PS data has arrived and copies the left channel until then
    // (aacsbr.c:1751-1758).  An LFE stays one channel (its type is not TYPE_SCE).
    // (A one-channel layout that signals SBR implicitly gets there in its first access unit: heaac_layout_dec_frame.)
    const int ps_sce = m4ac->sbr == 1 && m4ac->ps == 1;
    int outs = 0;
    for (int i = 0; i < layout->n_elements; i++)
        outs += ps_sce && layout->elem[i].type == HEAAC_ELEM_SCE ? 2 : layout->elem[i].channels;
    if (outs > HEAAC_MAX_PCM_PLANES) return NULL;
    const int most_outs = outs < 2 ? 2 : outs;
    HeaacLayoutDec *d = (HeaacLayoutDec *)calloc(1, sizeof(*d));
    if (!d) return NULL;
    d->dev = dev;
    d->m4ac = *m4ac;
    d->layout = *layout;
    d->n = layout->n_elements;
    d->out_channels = outs;
    d->tab = heaac_sbr_table_create(LAY_MAX_HDRS);
    d->h_coeffs = (float *)calloc((size_t)d->n * 2048, sizeof(float));
    d->h_tools = (HeaacToolsFrame *)calloc(d->n, sizeof(HeaacToolsFrame));
    HeaacPredictorState *ps = (HeaacPredictorState *)calloc(2 * HEAAC_MAX_PREDICTORS, sizeof(*ps));
    bool ok = d->tab && d->h_coeffs && d->h_tools && ps;
    if (ok) for (int i = 0; i < 2 * HEAAC_MAX_PREDICTORS; i++) ps[i].var0 = ps[i].var1 = 1.0f;   // reset_predict_state, :507-515
    for (int i = 0; ok && i < d->n; i++) {
        LayElem &e = d->e[i];
        e.channels = layout->elem[i].channels;
        e.cfg_lc = e.channels == 2 ? HEAAC_CFG_LC_STEREO : HEAAC_CFG_LC_MONO;
        e.cfg_he = e.channels == 2 ? HEAAC_CFG_HEV1 : HEAAC_CFG_HEV1_MONO;
        e.out_channels = e.channels;
        if (ps_sce && layout->elem[i].type == HEAAC_ELEM_SCE) { e.cfg_he = HEAAC_CFG_HEV2; e.out_channels = 2; }
        e.first_out = i ? d->e[i - 1].first_out + d->e[i - 1].out_channels : 0;
        heaac_sbr_stream_init(&e.sst, 1);
        ok = hipMalloc((void **)&e.d_coeffs, 2 * 1024 * 4) == hipSuccess &&
             hipMalloc((void **)&e.d_ics, 2 * sizeof(HeaacIcs)) == hipSuccess &&
             hipMalloc((void **)&e.d_tools, sizeof(HeaacToolsFrame)) == hipSuccess &&
             hipMalloc((void **)&e.d_state, LAY_STATE_WORDS * 4) == hipSuccess &&
             hipMalloc((void **)&e.d_ps, sizeof(HeaacPsFrame)) == hipSuccess &&
             hipMalloc((void **)&e.d_pred, 2 * HEAAC_MAX_PREDICTORS * sizeof(*ps)) == hipSuccess &&
             hipMalloc((void **)&e.d_sbr, sizeof(HeaacSbrFrame)) == hipSuccess &&
             hipMalloc((void **)&e.d_f32, 2 * 2048 * 4) == hipSuccess &&
             hipMemset(e.d_state, 0, LAY_STATE_WORDS * 4) == hipSuccess &&
             hipMemcpy(e.d_pred, ps, 2 * HEAAC_MAX_PREDICTORS * sizeof(*ps), hipMemcpyHostToDevice) == hipSuccess;
    }
    for (int id = 0; ok && id < 16; id++)
        if (layout->slot_of[HEAAC_ELEM_CCE][id] && !d->cpl) ok = (d->cpl = coupled_alloc(ps)) != NULL;
    free(ps);
    const int32_t seed = 0x1f2e3d4c;                                   // ac->random_state, aacdec.c:558
    for (int i = 0; i < LAY_MAX_HDRS; i++) d->hdr[i].kx = 32;          // kx' = 32, m = 0 (aacsbr.c:130)
    ok = ok && hipMalloc((void **)&d->d_hdr, sizeof(d->hdr)) == hipSuccess &&
         hipMalloc((void **)&d->d_rng, 4) == hipSuccess &&
         hipMalloc((void **)&d->d_pcm, (size_t)most_outs * 2048 * 2) == hipSuccess &&
         hipMemcpy(d->d_hdr, d->hdr, sizeof(d->hdr), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(d->d_rng, &seed, 4, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { heaac_layout_dec_destroy(d); return NULL; }
    return d;
}

int heaac_layout_dec_frame(HeaacLayoutDec *d, const uint8_t *buf, int size, void *data, int *data_size,
                           HeaacLayoutOut *out)
{
    if (!d || !buf || size < 2 || !data || !data_size) return -1;
    HeaacAacFrameInfo fi;
    // the parser works on copies of the window histories until the whole unit has parsed
    HeaacAacStream st[HEAAC_MAX_ELEMENTS];
    for (int i = 0; i < d->n; i++) st[i] = d->e[i].ast;
    LayCoupled *c = d->cpl;
    HeaacCceOut co = { NULL, NULL, NULL, NULL, NULL };
    if (c) { co.cce = &c->h_cce[0][0]; co.coeffs = &c->h_coeffs[0][0]; co.ics = c->h_ics; co.tools = c->h_tools; co.elem = c->h_elem; }
    if (heaac_aac_parse_frame_layout_ex(&d->m4ac, &d->layout, st, buf, size, d->h_coeffs, &d->h_ics[0][0], d->h_tools,
                                        d->h_elem, c ? &co : NULL, &fi) != HEAAC_PARSE_OK) {
        // No samples.  Where the refusal is the reference's own, what its element decoders had done by then stays
        // done (heaac_parse.h, HEAAC_REFUSED_*): the window histories the parser has moved, and -- through the records
        // it left for the elements marked present, in bitstream order -- the noise generator and the predictors.
        if (fi.refused & HEAAC_REFUSED_AS_REFERENCE)
            for (int i = 0; i < d->n; i++) d->e[i].ast = st[i];
        if (fi.refused & HEAAC_REFUSED_RUN_TOOLS) {
            const int main_profile = d->m4ac.object_type == HEAAC_AOT_AAC_MAIN;
            for (int seq = 0; seq < d->n; seq++)
                for (int i = 0; i < d->n; i++) {
                    if (!d->h_elem[i].present || d->h_elem[i].seq != seq) continue;
                    LayElem &e = d->e[i];
                    HeaacPredictorState *pr = main_profile ? e.d_pred : NULL;
                    if (hipMemcpy(e.d_coeffs, d->h_coeffs + (size_t)i * 2048, (size_t)e.channels * 4096, hipMemcpyHostToDevice) != hipSuccess ||
                        hipMemcpy(e.d_tools, &d->h_tools[i], sizeof(HeaacToolsFrame), hipMemcpyHostToDevice) != hipSuccess ||
                        heaac_spectral_tools_batch_ex(d->dev, e.channels, HEAAC_TOOLS_ALL, e.d_coeffs, e.d_tools, d->d_rng, d->d_rng,
                                                      pr, pr, NULL, NULL, 0, 1, NULL) != HEAAC_OK)
                        return -1;
                }
            (void)hipDeviceSynchronize();
        }
        return -1;
    }
    for (int i = 0; i < d->n; i++)
        if (!d->h_elem[i].present) return -1;
    // the coupling elements of this access unit: slot k of the layout's list, the same in every output slot's row
    int cce_here[HEAAC_MAX_CCE] = { 0 }, n_cce = 0;
    for (int k = 0; c && k < HEAAC_MAX_CCE; k++) {
        cce_here[k] = c->h_cce[0][k].present;
        n_cce += cce_here[k];
        if (c->seen[k] && !cce_here[k]) return -1;
    }
    for (int i = 0; i < d->n; i++) d->e[i].ast = st[i];
    for (int k = 0; k < HEAAC_MAX_CCE; k++)
        if (cce_here[k]) c->seen[k] = 1;
    // "Dependent coupling is not supported together with LTP" (apply_dependent_coupling :1822-1826 returns): an LTP
    // profile stream (an ADTS header can say so) keeps its coupling elements but nothing couples in the spectrum
    if (n_cce && d->m4ac.object_type == 4)
        for (int i = 0; i < d->n; i++)
            for (int k = 0; k < HEAAC_MAX_CCE; k++)
                if (c->h_cce[i][k].coupling_point != HEAAC_CC_AFTER_IMDCT) c->h_cce[i][k].n_links = 0;
    if (!d->locked) {
        // implicit SBR counts only when the first access unit carries it (aacdec.c:1666-1675)
        if (d->m4ac.sbr == -1) {
            d->m4ac.sbr = 0;
            for (int i = 0; i < d->n; i++)
                if (d->h_elem[i].sbr_payload_bit >= 0) d->m4ac.sbr = 1;
            for (int k = 0; c && k < HEAAC_MAX_CCE; k++)
                if (cce_here[k] && c->h_elem[k].sbr_payload_bit >= 0) d->m4ac.sbr = 1;
            // ... and in a stream of ONE channel the first payload turns Parametric Stereo on with it: the output is
            // configured again, now with two channels (decode_extension_payload, aacdec.c:1670-1673)
            if (d->m4ac.sbr == 1 && d->m4ac.ps == -1 && d->out_channels == 1 && d->n == 1 &&
                d->layout.elem[0].type == HEAAC_ELEM_SCE) {
                d->m4ac.ps = 1;
                d->e[0].cfg_he = HEAAC_CFG_HEV2;
                d->e[0].out_channels = d->out_channels = 2;
            }
        }
        d->locked = 1;
    }
    const int he = d->m4ac.sbr == 1;
    const int mode = he ? heaac_sbr_output_mode(&d->m4ac) : 0;        // 1: the output at the core rate (shim.hip)
    if (mode < 0) return -1;
    const int len = he && !mode ? 2048 : 1024;
    const int main_profile = d->m4ac.object_type == HEAAC_AOT_AAC_MAIN;
    // uploads, then the spectral tools of the elements in bitstream order (one noise generator)
    for (int i = 0; i < d->n; i++) {
        LayElem &e = d->e[i];
        if (hipMemcpy(e.d_coeffs, d->h_coeffs + (size_t)i * 2048, (size_t)e.channels * 4096, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(e.d_ics, d->h_ics[i], 2 * sizeof(HeaacIcs), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(e.d_tools, &d->h_tools[i], sizeof(HeaacToolsFrame), hipMemcpyHostToDevice) != hipSuccess)
            return -1;
    }
    if (n_cce &&
        (hipMemcpy(c->d_cce, c->h_cce, (size_t)d->n * HEAAC_MAX_CCE * sizeof(HeaacCceFrame), hipMemcpyHostToDevice) != hipSuccess ||
         hipMemcpy(c->d_coeffs, c->h_coeffs, sizeof(c->h_coeffs), hipMemcpyHostToDevice) != hipSuccess ||
         hipMemcpy(c->d_ics, c->h_ics, sizeof(c->h_ics), hipMemcpyHostToDevice) != hipSuccess ||
         hipMemcpy(c->d_tools, c->h_tools, sizeof(c->h_tools), hipMemcpyHostToDevice) != hipSuccess))
        return -1;
    // header records the SBR readers have added to the table since the last upload
    auto sync_headers = [&]() -> int {
        const size_t have = heaac_sbr_table_count(d->tab);
        if (have > d->hdr_uploaded) {
            memcpy(d->hdr + d->hdr_uploaded, heaac_sbr_table_data(d->tab) + d->hdr_uploaded,
                   (have - d->hdr_uploaded) * sizeof(HeaacSbrHeader));
            if (hipMemcpy(d->d_hdr + d->hdr_uploaded, d->hdr + d->hdr_uploaded,
                          (have - d->hdr_uploaded) * sizeof(HeaacSbrHeader), hipMemcpyHostToDevice) != hipSuccess)
                return -1;
            d->hdr_uploaded = have;
        }
        return 0;
    };
    // A coupling element's tools as a whole at its place in the stream (nothing couples INTO it); an output element's
    // first half there, its second half -- coupling, TNS, coupling -- once every coupling element is through.
    auto cce_tools = [&](int outputs_before) -> int {
        for (int seq = 0; seq < n_cce; seq++)
            for (int k = 0; k < HEAAC_MAX_CCE; k++) {
                const HeaacCceFrame &r = c->h_cce[0][k];
                if (!r.present || r.seq != seq || r.outputs_before != outputs_before) continue;
                HeaacPredictorState *pr = main_profile ? c->d_pred + k * HEAAC_MAX_PREDICTORS : NULL;
                if (heaac_spectral_tools_batch_ex(d->dev, 1, HEAAC_TOOLS_ALL, c->d_coeffs + k * 1024, c->d_tools + k,
                                                  d->d_rng, d->d_rng, pr, pr, NULL, NULL, 0, 1, NULL) != HEAAC_OK)
                    return -1;
            }
        return 0;
    };
    for (int seq = 0; seq < d->n; seq++) {
        if (n_cce && cce_tools(seq)) return -1;
        for (int i = 0; i < d->n; i++) {
            if (d->h_elem[i].seq != seq) continue;
            LayElem &e = d->e[i];
            HeaacPredictorState *pr = main_profile ? e.d_pred : NULL;
            if (heaac_spectral_tools_batch_ex(d->dev, e.channels, n_cce ? HEAAC_TOOLS_PRE : HEAAC_TOOLS_ALL, e.d_coeffs, e.d_tools,
                                              d->d_rng, d->d_rng, pr, pr, NULL, NULL, 0, 1, NULL) != HEAAC_OK)
                return -1;
        }
    }
    if (n_cce) {
        if (cce_tools(d->n)) return -1;
        for (int i = 0; i < d->n; i++) {
            LayElem &e = d->e[i];
            if (heaac_spectral_tools_batch_ex(d->dev, e.channels, HEAAC_TOOLS_POST, e.d_coeffs, e.d_tools, NULL, NULL, NULL, NULL,
                                              c->d_cce + (size_t)i * HEAAC_MAX_CCE, c->d_coeffs, HEAAC_MAX_CCE, 1, NULL) != HEAAC_OK)
                return -1;
        }
        // the coupling channels that couple behind the IMDCT: their own IMDCT -- and SBR, as a mono element's
        // (:1920-1927) -- first (type 2 before types 1 and 0).  A payload behind a coupling element that couples in
        // the spectrum is read all the same (decode_extension_payload does not look at the coupling point).
        for (int k = 0; k < HEAAC_MAX_CCE; k++) {
            if (!cce_here[k]) continue;
            const bool after = c->h_cce[0][k].coupling_point == HEAAC_CC_AFTER_IMDCT;
            float *st_k = c->d_state + (size_t)k * CCE_STATE_WORDS;
            if (!he) {
                if (after && heaac_lc_decode_batch(d->dev, 1, c->d_coeffs + k * 1024, c->d_ics + k, st_k, st_k, c->d_ret + k * 2048,
                                                   HEAAC_PCM_F32_PLANAR, 1, NULL) != HEAAC_OK)
                    return -1;
                continue;
            }
            HeaacSbrFrame sbr;
            const HeaacAacElementInfo &ei = c->h_elem[k];
            if (ei.sbr_payload_bit >= 0)
                (void)heaac_sbr_parse_payload(&c->sst[k], d->tab, d->m4ac.sample_rate, buf, size, ei.sbr_payload_bit, ei.sbr_payload_bytes,
                                              ei.sbr_crc, 1, ei.sbr_misplaced ? HEAAC_SBR_MISPLACED : 0, &sbr, NULL, NULL);
            else if (after)
                heaac_sbr_no_payload(&c->sst[k], 1, &sbr, NULL);
            if (!after) continue;
            if (sync_headers()) return -1;
            if (heaac_validate_frame(HEAAC_CFG_HEV1_MONO, &sbr, d->hdr, LAY_MAX_HDRS, NULL)) return -1;
            if (hipMemcpy(c->d_sbr + k, &sbr, sizeof(sbr), hipMemcpyHostToDevice) != hipSuccess) return -1;
            if (heaac_he_decode_batch_ex(d->dev, HEAAC_CFG_HEV1_MONO, mode ? HEAAC_HE_DOWNSAMPLED : 0, c->d_coeffs + k * 1024, c->d_ics + k,
                                         c->d_sbr + k, d->d_hdr, LAY_MAX_HDRS, NULL, st_k, st_k, c->d_ret + k * 2048,
                                         HEAAC_PCM_F32_PLANAR, 1, NULL) != HEAAC_OK)
                return -1;
        }
    }
    HeaacPlaneRef planes[HEAAC_MAX_PCM_PLANES];
    for (int i = 0; i < d->n; i++) {
        LayElem &e = d->e[i];
        int rc;
        if (!he) {
            rc = heaac_lc_decode_batch(d->dev, e.channels, e.d_coeffs, e.d_ics, e.d_state, e.d_state, e.d_f32,
                                       HEAAC_PCM_F32_PLANAR, 1, NULL);
        } else {
            HeaacSbrFrame sbr;
            HeaacPsFrame ps;
            const bool with_ps = e.cfg_he == HEAAC_CFG_HEV2;
            const HeaacAacElementInfo &ei = d->h_elem[i];
            if (ei.sbr_payload_bit >= 0) {
                // a failed payload leaves its degraded record (start = 0) and the decode goes on, as ff_sbr_apply does
                (void)heaac_sbr_parse_payload(&e.sst, d->tab, d->m4ac.sample_rate, buf, size, ei.sbr_payload_bit,
                                              ei.sbr_payload_bytes, ei.sbr_crc, e.channels,
                                              (ei.sbr_misplaced ? HEAAC_SBR_MISPLACED : 0) | (with_ps ? HEAAC_SBR_ALLOW_PS : 0),
                                              &sbr, with_ps ? &ps : NULL, NULL);
            } else {
                heaac_sbr_no_payload(&e.sst, e.channels, &sbr, with_ps ? &ps : NULL);
            }
            if (sync_headers()) return -1;
            if (heaac_validate_frame(e.cfg_he, &sbr, d->hdr, LAY_MAX_HDRS, with_ps ? &ps : NULL)) return -1;
            if (hipMemcpy(e.d_sbr, &sbr, sizeof(sbr), hipMemcpyHostToDevice) != hipSuccess) return -1;
            if (with_ps && hipMemcpy(e.d_ps, &ps, sizeof(ps), hipMemcpyHostToDevice) != hipSuccess) return -1;
            rc = heaac_he_decode_batch_ex(d->dev, e.cfg_he, mode ? HEAAC_HE_DOWNSAMPLED : 0, e.d_coeffs, e.d_ics, e.d_sbr, d->d_hdr,
                                          LAY_MAX_HDRS, with_ps ? e.d_ps : NULL, e.d_state, e.d_state, e.d_f32, HEAAC_PCM_F32_PLANAR,
                                          1, NULL);
        }
        if (rc != HEAAC_OK) return -1;
        // every AFTER_IMDCT element in tag order, every gain list it lands on this element (apply_channel_coupling
        // :1870-1898; apply_independent_coupling :1849-1862 over 1024 << sbr samples): one target channel at a time,
        // its plane and the coupling channel's as len / 1024 "frames" of the batched op
        for (int k = 0; k < HEAAC_MAX_CCE; k++) {
            if (!cce_here[k] || c->h_cce[i][k].coupling_point != HEAAC_CC_AFTER_IMDCT) continue;
            for (int l = 0; l < c->h_cce[i][k].n_links; l++) {
                HeaacCoupling g[2];
                memset(g, 0, sizeof(g));
                g[0].on[0] = g[1].on[0] = 1;
                g[0].gain[0] = g[1].gain[0] = c->h_cce[i][k].link[l].gain[0];
                if (hipMemcpy(c->d_gain, g, sizeof(g), hipMemcpyHostToDevice) != hipSuccess) return -1;
                if (heaac_couple_after_imdct_batch(d->dev, 1, e.d_f32 + (size_t)c->h_cce[i][k].link[l].target_ch * len, c->d_ret + k * 2048,
                                                   c->d_gain, NULL, len / 1024, NULL) != HEAAC_OK)
                    return -1;
            }
        }
        const int np = he ? e.out_channels : e.channels;
        for (int c = 0; c < np; c++) {
            planes[e.first_out + c].d_base = e.d_f32 + (size_t)c * len;
            planes[e.first_out + c].frame_stride = (size_t)np * len;
        }
    }
    if (heaac_pcm_interleave_batch(d->dev, d->out_channels, planes, len, HEAAC_PCM_S16_INTERLEAVED, d->d_pcm, 1, NULL) != HEAAC_OK)
        return -1;
    const int bytes = len * d->out_channels * 2;
    if (*data_size < bytes) return -1;                       // "Output buffer too small" (aacdec.c:2087-2092)
    if (hipMemcpy(data, d->d_pcm, bytes, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    *data_size = bytes;
    if (out) {
        out->channels = d->out_channels;
        out->channel_layout = d->layout.channel_layout;
        out->frame_size = len;
        out->sample_rate = he && !mode ? 2 * d->m4ac.sample_rate : d->m4ac.sample_rate;
    }
    // aacdec.c:2102-2107: bytes consumed, or the whole packet when only zero padding follows
    const int consumed = (fi.bits_consumed + 7) >> 3;
    int off = consumed;
    while (off < size && !buf[off]) off++;
    return size > off ? consumed : size;
}
