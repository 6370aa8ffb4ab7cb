/* or_tools.c -- CPU restatement of the spectral tools that precede the IMDCT
 * (SURVEY.md s8f N1).  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * PARITY UNPINNED by the reference: its tree holds no test or vector for these
 * functions; tests/test_oracle_props.py checks domain properties (M/S
 * butterflies are 2x an involution, TNS with a zero filter is the identity,
 * the all-pole TNS filter undoes the matching FIR, intensity bands are exact
 * scaled copies).
 */
#include <string.h>
#include <math.h>
#include "oracle.h"

/* butterflies_float_c, dsputil.c:3899-3908 */
static void butterflies_float(float *v1, float *v2, int len)
{
    for (int i = 0; i < len; i++) {
        float t = v1[i] - v2[i];
        v1[i] += v2[i];
        v2[i] = t;
    }
}

/* apply_mid_side_stereo, aacdec.c:1390-1411 */
static void or_mid_side(const HeaacToolsFrame *t, float *ch0, float *ch1)
{
    const HeaacToolsIcs *ics = &t->ch[0].ics;
    int idx = 0;
    for (int g = 0; g < ics->num_window_groups; g++) {
        for (int i = 0; i < ics->max_sfb; i++, idx++) {
            if (t->ms_mask[idx] &&
                t->ch[0].band_type[idx] < HEAAC_NOISE_BT && t->ch[1].band_type[idx] < HEAAC_NOISE_BT) {
                for (int group = 0; group < ics->group_len[g]; group++)
                    butterflies_float(ch0 + group * 128 + ics->swb_offset[i],
                                      ch1 + group * 128 + ics->swb_offset[i],
                                      ics->swb_offset[i + 1] - ics->swb_offset[i]);
            }
        }
        ch0 += ics->group_len[g] * 128;
        ch1 += ics->group_len[g] * 128;
    }
}

/* apply_intensity_stereo, aacdec.c:1420-1451.  The reference skips whole runs of equal
 * band type (band_type_run_end); testing each band visits the same bands. */
static void or_intensity(const HeaacToolsFrame *t, float *coef0, float *coef1)
{
    const HeaacToolsIcs *ics = &t->ch[1].ics;
    const HeaacToolsChannel *sce1 = &t->ch[1];
    int idx = 0;
    for (int g = 0; g < ics->num_window_groups; g++) {
        for (int i = 0; i < ics->max_sfb; i++, idx++) {
            if (sce1->band_type[idx] == HEAAC_INTENSITY_BT || sce1->band_type[idx] == HEAAC_INTENSITY_BT2) {
                int c = -1 + 2 * (sce1->band_type[idx] - 14);
                if (t->ms_present)
                    c *= 1 - 2 * t->ms_mask[idx];
                const float scale = c * sce1->sf[idx];
                for (int group = 0; group < ics->group_len[g]; group++)
                    for (int k = ics->swb_offset[i]; k < ics->swb_offset[i + 1]; k++)
                        coef1[group * 128 + k] = scale * coef0[group * 128 + k];
            }
        }
        coef0 += ics->group_len[g] * 128;
        coef1 += ics->group_len[g] * 128;
    }
}

/* compute_lpc_coefs(autoc, order, lpc, 0, 0, 0), lpc.h:61-103 with LPC_TYPE float */
static void or_lpc_coefs(const float *autoc, int max_order, float *lpc)
{
    for (int i = 0; i < max_order; i++) {
        float r = -autoc[i];
        lpc[i] = r;
        for (int j = 0; j < (i + 1) >> 1; j++) {
            float f = lpc[j];
            float b = lpc[i - 1 - j];
            lpc[j]         = f + r * b;
            lpc[i - 1 - j] = b + r * f;
        }
    }
}

#define OR_MIN(a, b) ((a) < (b) ? (a) : (b))
#define OR_MAX(a, b) ((a) > (b) ? (a) : (b))

/* apply_tns(coef, tns, ics, decode = 1), aacdec.c:1698-1736 */
static void or_tns(float coef[1024], const HeaacTns *tns, const HeaacToolsIcs *ics)
{
    const int mmm = OR_MIN(ics->tns_max_bands, ics->max_sfb);
    float lpc[HEAAC_TNS_MAX_ORDER];
    for (int w = 0; w < ics->num_windows; w++) {
        int bottom = ics->num_swb;
        for (int filt = 0; filt < tns->n_filt[w]; filt++) {
            const int top = bottom;
            bottom = OR_MAX(0, top - tns->length[w][filt]);
            const int order = tns->order[w][filt];
            if (order == 0)
                continue;
            or_lpc_coefs(tns->coef[w][filt], order, lpc);
            int start = ics->swb_offset[OR_MIN(bottom, mmm)];
            const int end = ics->swb_offset[OR_MIN(top, mmm)];
            const int size = end - start;
            int inc;
            if (size <= 0)
                continue;
            if (tns->direction[w][filt]) {
                inc = -1;
                start = end - 1;
            } else {
                inc = 1;
            }
            start += w * 128;
            for (int m = 0; m < size; m++, start += inc)
                for (int i = 1; i <= OR_MIN(m, order); i++)
                    coef[start] -= coef[start - i * inc] * lpc[i - 1];
        }
    }
}

/* lcg_random, aacdec.c:502-505 (int arithmetic that wraps) */
static int lcg_random(int previous_val)
{
    return (int)((unsigned)previous_val * 1664525u + 1013904223u);
}

/* the NOISE_BT branch of decode_spectrum_and_dequant, aacdec.c:1003-1029 */
static int or_pns(const HeaacToolsChannel *ch, float *coef, int random_state)
{
    const HeaacToolsIcs *ics = &ch->ics;
    int idx = 0;
    for (int g = 0; g < ics->num_window_groups; g++) {
        const int g_len = ics->group_len[g];
        for (int i = 0; i < ics->max_sfb; i++, idx++) {
            if (ch->band_type[idx] != HEAAC_NOISE_BT)
                continue;
            float *cfo = coef + ics->swb_offset[i];
            const int off_len = ics->swb_offset[i + 1] - ics->swb_offset[i];
            for (int group = 0; group < g_len; group++, cfo += 128) {
                for (int k = 0; k < off_len; k++) {
                    random_state = lcg_random(random_state);
                    cfo[k] = random_state;
                }
                float band_energy = 0.0;                      /* scalarproduct_float_c, dsputil.c:3910-3919 */
                for (int k = 0; k < off_len; k++)
                    band_energy += cfo[k] * cfo[k];
                const float scale = ch->sf[idx] / sqrtf(band_energy);
                for (int k = 0; k < off_len; k++)                 /* vector_fmul_scalar_c */
                    cfo[k] = cfo[k] * scale;
            }
        }
        coef += g_len << 7;
    }
    return random_state;
}

/* flt16_round / flt16_even / flt16_trunc, aacdec.c:1247-1269 (flt16_even's `& 0x00010000U >> 16`
 * parses as `& 1`: kept) */
static float flt16_round(float pf)
{
    union { float f; uint32_t i; } t; t.f = pf;
    t.i = (t.i + 0x00008000U) & 0xFFFF0000U;
    return t.f;
}
static float flt16_even(float pf)
{
    union { float f; uint32_t i; } t; t.f = pf;
    t.i = (t.i + 0x00007FFFU + (t.i & 0x00010000U >> 16)) & 0xFFFF0000U;
    return t.f;
}
static float flt16_trunc(float pf)
{
    union { float f; uint32_t i; } t; t.f = pf;
    t.i &= 0xFFFF0000U;
    return t.f;
}

/* predict, aacdec.c:1271-1297 (0.5 is a double literal there: the var updates are summed in double) */
static void or_predict(HeaacPredictorState *ps, float *coef, int output_enable)
{
    const float sf_scale = HEAAC_SF_SCALE;
    const float a     = 0.953125; // 61.0 / 64
    const float alpha = 0.90625;  // 29.0 / 32
    float e0, e1;
    float pv;
    float k1, k2;

    k1 = ps->var0 > 1 ? ps->cor0 * flt16_even(a / ps->var0) : 0;
    k2 = ps->var1 > 1 ? ps->cor1 * flt16_even(a / ps->var1) : 0;

    pv = flt16_round(k1 * ps->r0 + k2 * ps->r1);
    if (output_enable)
        *coef += pv * sf_scale;

    e0 = *coef / sf_scale;
    e1 = e0 - k1 * ps->r0;

    ps->cor1 = flt16_trunc(alpha * ps->cor1 + ps->r1 * e1);
    ps->var1 = flt16_trunc(alpha * ps->var1 + 0.5 * (ps->r1 * ps->r1 + e1 * e1));
    ps->cor0 = flt16_trunc(alpha * ps->cor0 + ps->r0 * e0);
    ps->var0 = flt16_trunc(alpha * ps->var0 + 0.5 * (ps->r0 * ps->r0 + e0 * e0));

    ps->r1 = flt16_trunc(a * (ps->r0 - k1 * e0));
    ps->r0 = flt16_trunc(a * e0);
}

static void or_reset_predict_state(HeaacPredictorState *ps)
{
    ps->r0 = 0.0f; ps->r1 = 0.0f; ps->cor0 = 0.0f; ps->cor1 = 0.0f; ps->var0 = 1.0f; ps->var1 = 1.0f;
}

/* apply_prediction, aacdec.c:1302-1322 (predictor_initialized: the caller's initial state) */
static void or_prediction(const HeaacToolsChannel *ch, float *coef, HeaacPredictorState *st)
{
    if (ch->ics.num_windows != 8) {
        for (int sfb = 0; sfb < ch->pred.pred_sfb_max; sfb++)
            for (int k = ch->ics.swb_offset[sfb]; k < ch->ics.swb_offset[sfb + 1]; k++)
                or_predict(&st[k], &coef[k], ch->pred.predictor_present && ch->pred.prediction_used[sfb]);
        if (ch->pred.predictor_reset_group)
            for (int i = ch->pred.predictor_reset_group - 1; i < HEAAC_MAX_PREDICTORS; i += 30)
                or_reset_predict_state(&st[i]);
    } else {
        for (int i = 0; i < HEAAC_MAX_PREDICTORS; i++)
            or_reset_predict_state(&st[i]);
    }
}

/* apply_dependent_coupling (aacdec.c:1813-1843): dest += gain[idx] * src over the coupling channel's
 * non-zero bands, band offsets and grouping of the COUPLING channel. */
static void or_dependent_coupling(float *dest, const float *src, const HeaacCceFrame *cce, const float *gain_list)
{
    const HeaacToolsIcs *ics = &cce->ics;
    const uint16_t *offsets = ics->swb_offset;
    int idx = 0;
    for (int g = 0; g < ics->num_window_groups; g++) {
        for (int i = 0; i < ics->max_sfb; i++, idx++) {
            if (cce->band_type[idx] != 0) {                            /* ZERO_BT */
                const float gain = gain_list[idx];
                for (int group = 0; group < ics->group_len[g]; group++)
                    for (int k = offsets[i]; k < offsets[i + 1]; k++)
                        dest[group * 128 + k] += gain * src[group * 128 + k];
            }
        }
        dest += ics->group_len[g] * 128;
        src  += ics->group_len[g] * 128;
    }
}

/* apply_channel_coupling (aacdec.c:1870-1898) for one coupling point: the coupling elements in ascending tag order
 * (the slots' order), each gain list that lands on the target element (resolved by the parser into links) */
static void or_channel_coupling(int channels, float *c0, float *c1, const HeaacCceFrame *cce, const float *cce_coeffs,
                                int n_cce, int point)
{
    for (int e = 0; e < n_cce; e++) {
        if (!cce[e].present || cce[e].coupling_point != point) continue;
        for (int l = 0; l < cce[e].n_links; l++) {
            const HeaacCceLink *k = &cce[e].link[l];
            if (k->target_ch >= channels) continue;
            or_dependent_coupling(k->target_ch ? c1 : c0, cce_coeffs + (size_t)e * 1024, &cce[e], k->gain);
        }
    }
}

/* decode_cpe's tail (aacdec.c:1483-1492) + spectral_to_sample's TNS calls (:1913-1916);
 * rng_in != NULL: noise substitution first, channel 0 then channel 1 (decode_ics order) */
void oracle_spectral_tools_batch(int channels, float *coeffs, const HeaacToolsFrame *tools,
                                 const int32_t *rng_in, int32_t *rng_out,
                                 const HeaacPredictorState *pred_in, HeaacPredictorState *pred_out, size_t n)
{
    oracle_spectral_tools_batch_ex(channels, HEAAC_TOOLS_ALL, coeffs, tools, rng_in, rng_out, pred_in, pred_out,
                                   NULL, NULL, 0, n);
}

/* The same in the two halves the reference runs at different times: PRE = what decode_ics / decode_cpe do while an
 * element is parsed (noise substitution, prediction, M/S, intensity), POST = what spectral_to_sample does to the
 * element before its IMDCT (aacdec.c:1911-1918): dependent coupling at BEFORE_TNS, TNS, dependent coupling at
 * BETWEEN_TNS_AND_IMDCT.  cce [n][n_cce], cce_coeffs [n][n_cce][1024]: the access unit's coupling elements after
 * their own tools. */
void oracle_spectral_tools_batch_ex(int channels, int stages, float *coeffs, const HeaacToolsFrame *tools,
                                    const int32_t *rng_in, int32_t *rng_out,
                                    const HeaacPredictorState *pred_in, HeaacPredictorState *pred_out,
                                    const HeaacCceFrame *cce, const float *cce_coeffs, int n_cce, size_t n)
{
    const int pre = stages & HEAAC_TOOLS_PRE, post = stages & HEAAC_TOOLS_POST;
    if (!pre) { rng_in = NULL; pred_in = NULL; }
    if (pred_in && pred_out != pred_in)
        memcpy(pred_out, pred_in, n * (size_t)channels * HEAAC_MAX_PREDICTORS * sizeof(*pred_in));
    for (size_t f = 0; f < n; f++) {
        const HeaacToolsFrame *t = &tools[f];
        float *c0 = coeffs + f * (size_t)channels * 1024, *c1 = c0 + 1024;
        if (rng_in) {
            int rs = rng_in[f];
            for (int c = 0; c < channels; c++)
                rs = or_pns(&t->ch[c], c ? c1 : c0, rs);
            rng_out[f] = rs;
        }
        HeaacPredictorState *p0 = pred_in ? pred_out + f * (size_t)channels * HEAAC_MAX_PREDICTORS : NULL;
        HeaacPredictorState *p1 = p0 ? p0 + HEAAC_MAX_PREDICTORS : NULL;
        const int common = channels == 2 && t->common_window;
        if (p0 && !common) {                       /* decode_ics, aacdec.c:1381-1382 */
            or_prediction(&t->ch[0], c0, p0);
            if (channels == 2) or_prediction(&t->ch[1], c1, p1);
        }
        if (channels == 2 && pre) {
            if (t->common_window && t->ms_present)
                or_mid_side(t, c0, c1);
            if (p0 && common) {                    /* decode_cpe, aacdec.c:1486-1489 */
                or_prediction(&t->ch[0], c0, p0);
                or_prediction(&t->ch[1], c1, p1);
            }
            or_intensity(t, c0, c1);
        }
        if (!post) continue;
        if (n_cce)
            or_channel_coupling(channels, c0, c1, cce + f * (size_t)n_cce, cce_coeffs + f * (size_t)n_cce * 1024, n_cce,
                                HEAAC_CC_BEFORE_TNS);
        if (t->ch[0].tns.present)
            or_tns(c0, &t->ch[0].tns, &t->ch[0].ics);
        if (channels == 2 && t->ch[1].tns.present)
            or_tns(c1, &t->ch[1].tns, &t->ch[1].ics);
        if (n_cce)
            or_channel_coupling(channels, c0, c1, cce + f * (size_t)n_cce, cce_coeffs + f * (size_t)n_cce * 1024, n_cce,
                                HEAAC_CC_BETWEEN_TNS_AND_IMDCT);
    }
}
