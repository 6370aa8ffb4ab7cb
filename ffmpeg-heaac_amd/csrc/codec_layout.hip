// codec_layout.hip -- aac_decode_frame for multi-element layouts behind the AVCodec surface (shim.hip):
// the element loop is heaac_aac_parse_frame_layout (aac_parse.c), spectral_to_sample (aacdec.c:1903-1933) is one
// decode call per element on that element's own state record, float_to_int16_interleave over output_data[]
// (:2096-2097) is heaac_pcm_interleave_batch over the elements' float planes in layout order.
//   * one noise generator for the stream, run through the elements in bitstream order (decode_spectrum_and_dequant
//     draws from ac->random_state as it parses, :1049-1054);
//   * SBR per element (che->sbr): an element's payload is the fill element directly behind it; once the stream has
//     SBR (explicitly, or implicitly by a payload in the FIRST access unit, :1666-1675) every element goes through
//     ff_sbr_apply, with a start = 0 record ("pure upsampling") where it has no payload -- an LFE never has one;
//   * an access unit that leaves an element of the layout out is refused: the reference transforms whatever that
//     element's buffers still hold from an earlier frame, which no record of this path carries.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include "codec_layout.h"

#define LAY_MAX_HDRS 64

struct LayElem {
    int cfg_lc, cfg_he, channels;
    HeaacAacStream ast;
    HeaacSbrStream sst;
    float *d_coeffs;              // [2][1024]
    HeaacIcs *d_ics;              // [2]
    HeaacToolsFrame *d_tools;
    float *d_state;               // HEAAC_STATE_WORDS_HEV1 (the largest of the four configurations)
    HeaacPredictorState *d_pred;  // [2][672]
    HeaacSbrFrame *d_sbr;
    float *d_f32;                 // [2][2048]
};

struct HeaacLayoutDec {
    HeaacDevice *dev;
    HeaacAacConfig m4ac;
    HeaacAacLayout layout;
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
};

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
    }
    if (d->d_hdr) (void)hipFree(d->d_hdr);
    if (d->d_rng) (void)hipFree(d->d_rng);
    if (d->d_pcm) (void)hipFree(d->d_pcm);
    heaac_sbr_table_destroy(d->tab);
    free(d->h_coeffs);
    free(d->h_tools);
    free(d);
}

HeaacLayoutDec *heaac_layout_dec_create(HeaacDevice *dev, const HeaacAacConfig *m4ac, const HeaacAacLayout *layout)
{
    if (!dev || !m4ac || !layout || layout->n_elements < 1 || layout->n_elements > HEAAC_MAX_ELEMENTS ||
        layout->channels < 1 || layout->channels > HEAAC_MAX_PCM_PLANES)
        return NULL;
    HeaacLayoutDec *d = (HeaacLayoutDec *)calloc(1, sizeof(*d));
    if (!d) return NULL;
    d->dev = dev;
    d->m4ac = *m4ac;
    d->layout = *layout;
    d->n = layout->n_elements;
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
        heaac_sbr_stream_init(&e.sst, 1);
        ok = hipMalloc((void **)&e.d_coeffs, 2 * 1024 * 4) == hipSuccess &&
             hipMalloc((void **)&e.d_ics, 2 * sizeof(HeaacIcs)) == hipSuccess &&
             hipMalloc((void **)&e.d_tools, sizeof(HeaacToolsFrame)) == hipSuccess &&
             hipMalloc((void **)&e.d_state, HEAAC_STATE_WORDS_HEV1 * 4) == hipSuccess &&
             hipMalloc((void **)&e.d_pred, 2 * HEAAC_MAX_PREDICTORS * sizeof(*ps)) == hipSuccess &&
             hipMalloc((void **)&e.d_sbr, sizeof(HeaacSbrFrame)) == hipSuccess &&
             hipMalloc((void **)&e.d_f32, 2 * 2048 * 4) == hipSuccess &&
             hipMemset(e.d_state, 0, HEAAC_STATE_WORDS_HEV1 * 4) == hipSuccess &&
             hipMemcpy(e.d_pred, ps, 2 * HEAAC_MAX_PREDICTORS * sizeof(*ps), hipMemcpyHostToDevice) == hipSuccess;
    }
    free(ps);
    const int32_t seed = 0x1f2e3d4c;                                   // ac->random_state, aacdec.c:558
    for (int i = 0; i < LAY_MAX_HDRS; i++) d->hdr[i].kx = 32;          // kx' = 32, m = 0 (aacsbr.c:130)
    ok = ok && hipMalloc((void **)&d->d_hdr, sizeof(d->hdr)) == hipSuccess &&
         hipMalloc((void **)&d->d_rng, 4) == hipSuccess &&
         hipMalloc((void **)&d->d_pcm, (size_t)layout->channels * 2048 * 2) == hipSuccess &&
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
    if (heaac_aac_parse_frame_layout(&d->m4ac, &d->layout, st, buf, size, d->h_coeffs, &d->h_ics[0][0], d->h_tools,
                                     d->h_elem, &fi) != HEAAC_PARSE_OK)
        return -1;
    for (int i = 0; i < d->n; i++)
        if (!d->h_elem[i].present) return -1;
    for (int i = 0; i < d->n; i++) d->e[i].ast = st[i];
    if (!d->locked) {
        // implicit SBR counts only when the first access unit carries it (aacdec.c:1666-1675)
        if (d->m4ac.sbr == -1) {
            d->m4ac.sbr = 0;
            for (int i = 0; i < d->n; i++)
                if (d->h_elem[i].sbr_payload_bit >= 0) d->m4ac.sbr = 1;
        }
        d->locked = 1;
    }
    const int he = d->m4ac.sbr == 1;
    const int len = he ? 2048 : 1024;
    const int main_profile = d->m4ac.object_type == HEAAC_AOT_AAC_MAIN;
    // uploads, then the spectral tools of the elements in bitstream order (one noise generator)
    for (int i = 0; i < d->n; i++) {
        LayElem &e = d->e[i];
        if (hipMemcpy(e.d_coeffs, d->h_coeffs + (size_t)i * 2048, (size_t)e.channels * 4096, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(e.d_ics, d->h_ics[i], 2 * sizeof(HeaacIcs), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(e.d_tools, &d->h_tools[i], sizeof(HeaacToolsFrame), hipMemcpyHostToDevice) != hipSuccess)
            return -1;
    }
    for (int seq = 0; seq < d->n; seq++)
        for (int i = 0; i < d->n; i++) {
            if (d->h_elem[i].seq != seq) continue;
            LayElem &e = d->e[i];
            HeaacPredictorState *pr = main_profile ? e.d_pred : NULL;
            if (heaac_spectral_tools_batch(d->dev, e.channels, e.d_coeffs, e.d_tools, d->d_rng, d->d_rng, pr, pr, 1, NULL) != HEAAC_OK)
                return -1;
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
            const HeaacAacElementInfo &ei = d->h_elem[i];
            if (ei.sbr_payload_bit >= 0) {
                // a failed payload leaves its degraded record (start = 0) and the decode goes on, as ff_sbr_apply does
                (void)heaac_sbr_parse_payload(&e.sst, d->tab, d->m4ac.sample_rate, buf, size, ei.sbr_payload_bit,
                                              ei.sbr_payload_bytes, ei.sbr_crc, e.channels, 0, &sbr, NULL, NULL);
            } else {
                heaac_sbr_no_payload(&e.sst, e.channels, &sbr, NULL);
            }
            const size_t have = heaac_sbr_table_count(d->tab);
            if (have > d->hdr_uploaded) {
                memcpy(d->hdr + d->hdr_uploaded, heaac_sbr_table_data(d->tab) + d->hdr_uploaded,
                       (have - d->hdr_uploaded) * sizeof(HeaacSbrHeader));
                if (hipMemcpy(d->d_hdr + d->hdr_uploaded, d->hdr + d->hdr_uploaded,
                              (have - d->hdr_uploaded) * sizeof(HeaacSbrHeader), hipMemcpyHostToDevice) != hipSuccess)
                    return -1;
                d->hdr_uploaded = have;
            }
            if (heaac_validate_frame(e.cfg_he, &sbr, d->hdr, LAY_MAX_HDRS, NULL)) return -1;
            if (hipMemcpy(e.d_sbr, &sbr, sizeof(sbr), hipMemcpyHostToDevice) != hipSuccess) return -1;
            rc = heaac_he_decode_batch(d->dev, e.cfg_he, e.d_coeffs, e.d_ics, e.d_sbr, d->d_hdr, LAY_MAX_HDRS, NULL,
                                       e.d_state, e.d_state, e.d_f32, HEAAC_PCM_F32_PLANAR, 1, NULL);
        }
        if (rc != HEAAC_OK) return -1;
        for (int c = 0; c < e.channels; c++) {
            planes[d->layout.elem[i].first_channel + c].d_base = e.d_f32 + (size_t)c * len;
            planes[d->layout.elem[i].first_channel + c].frame_stride = (size_t)e.channels * len;
        }
    }
    if (heaac_pcm_interleave_batch(d->dev, d->layout.channels, planes, len, HEAAC_PCM_S16_INTERLEAVED, d->d_pcm, 1, NULL) != HEAAC_OK)
        return -1;
    const int bytes = len * d->layout.channels * 2;
    if (hipMemcpy(data, d->d_pcm, bytes, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    *data_size = bytes;
    if (out) {
        out->channels = d->layout.channels;
        out->channel_layout = d->layout.channel_layout;
        out->frame_size = len;
        out->sample_rate = he ? 2 * d->m4ac.sample_rate : d->m4ac.sample_rate;
    }
    // aacdec.c:2102-2107: bytes consumed, or the whole packet when only zero padding follows
    const int consumed = (fi.bits_consumed + 7) >> 3;
    int off = consumed;
    while (off < size && !buf[off]) off++;
    return size > off ? consumed : size;
}
