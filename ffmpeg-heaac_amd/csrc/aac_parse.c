/* aac_parse.c -- host-side AAC access-unit parser (include/heaac_parse.h).
 *
 * Own structure: a bit reader over the access unit, binary code trees built once from the ISO code /
 * length tables (aac_iso_tables.h), one pass per element that writes straight into the records of the
 * batched GPU entry points.  The VALUES follow the reference bit for bit: which bits are read in which
 * order (ISO/IEC 14496-3 tables 4.4 - 4.54 as aacdec.c reads them) and how a quantised line becomes a
 * float (decode_spectrum_and_dequant, aacdec.c:988-1245): mag(q) = q^(4/3) as a float for q < 16,
 * cbrtf(n) * n for an escape value, times the band's scalefactor -2^((sf - 200) / 4) with the line's
 * sign; pulses re-quantise the line as :1222-1236 does.
 */
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "heaac_parse.h"
#include "aac_iso_tables.h"
#include "parse_bits.h"

static Tree g_sf_tree, g_spec_tree[11];
static float g_pow2sf[428];                           /* ff_aac_pow2sf_tab: 2^((i - 200) / 4), aac_tablegen.h */
static float g_mag[16];                               /* q^(4/3), q = 0..15 (aactab.c: codebook vector values) */
static pthread_once_t g_once = PTHREAD_ONCE_INIT;
static int g_tables_bad;                              /* a code book did not fit its tree (never with the shipped tables) */

static void tables_init(void)
{
    g_tables_bad |= tree_build(&g_sf_tree, aac_sf_code, NULL, aac_sf_bits, 121);
    for (int b = 0; b < 11; b++)
        g_tables_bad |= tree_build(&g_spec_tree[b], NULL, aac_spec_code + aac_spec_first[b], aac_spec_bits + aac_spec_first[b],
                   aac_spec_first[b + 1] - aac_spec_first[b]);
    for (int i = 0; i < 428; i++) g_pow2sf[i] = (float)pow(2, (i - 200) / 4.);
    for (int q = 0; q < 16; q++) g_mag[q] = (float)pow((double)q, 4.0 / 3.0);
}

uint64_t heaac_aac_tables_fingerprint(void)
{
    uint64_t h = 1469598103934665603ull;
#define MIX(arr) do { const uint8_t *p_ = (const uint8_t *)(arr); for (size_t i_ = 0; i_ < sizeof(arr); i_++) { h ^= p_[i_]; h *= 1099511628211ull; } } while (0)
    MIX(aac_sf_code); MIX(aac_sf_bits); MIX(aac_spec_first); MIX(aac_spec_code); MIX(aac_spec_bits);
    MIX(aac_num_swb_1024); MIX(aac_num_swb_128); MIX(aac_pred_sfb_max); MIX(aac_tns_max_bands_1024);
    MIX(aac_tns_max_bands_128); MIX(aac_swb_offset_1024); MIX(aac_swb_offset_128); MIX(aac_tns_map);
#undef MIX
    return h;
}

/* ------------------------------------------------------------------------------------------ */
/* AudioSpecificConfig, ADTS                                                                     */
/* ------------------------------------------------------------------------------------------ */
static const int k_rates[16] = { 96000, 88200, 64000, 48000, 44100, 32000, 24000, 22050, 16000, 12000,
                                 11025, 8000, 7350, 0, 0, 0 };

static int get_object_type(Bits *b)
{
    int t = (int)bits(b, 5);
    if (t == 31) t = 32 + (int)bits(b, 6);
    return t;
}
static int get_sample_rate(Bits *b, int *index)
{
    *index = (int)bits(b, 4);
    return *index == 0x0f ? (int)bits(b, 24) : k_rates[*index];
}

int heaac_asc_parse(HeaacAacConfig *c, const uint8_t *buf, int size)
{
    if (!c || !buf || size <= 0) return HEAAC_PARSE_ERR_ARG;
    Bits b;
    bits_init(&b, buf, size);
    memset(c, 0, sizeof(*c));
    c->object_type = get_object_type(&b);
    c->sample_rate = get_sample_rate(&b, &c->sampling_index);
    c->chan_config = (int)bits(&b, 4);
    c->sbr = -1;
    c->ps = -1;
    if (c->object_type == HEAAC_AOT_SBR ||
        (c->object_type == HEAAC_AOT_PS && !((peek(&b, 3) & 0x03) && !(peek(&b, 9) & 0x3F)))) {
        if (c->object_type == HEAAC_AOT_PS) c->ps = 1;
        c->ext_object_type = HEAAC_AOT_SBR;
        c->sbr = 1;
        c->ext_sample_rate = get_sample_rate(&b, &c->ext_sampling_index);
        c->object_type = get_object_type(&b);
        if (c->object_type == 22)                      /* AOT_ER_BSAC */
            c->ext_chan_config = (int)bits(&b, 4);
    }
    const int specific = b.pos;
    if (c->object_type == 36)                          /* AOT_ALS: not an AAC configuration */
        return HEAAC_PARSE_ERR_UNSUPPORTED;
    if (c->ext_object_type != HEAAC_AOT_SBR) {
        while (bits_left(&b) > 15) {
            if (peek(&b, 11) == 0x2b7) {               /* sync extension */
                bits(&b, 11);
                c->ext_object_type = get_object_type(&b);
                if (c->ext_object_type == HEAAC_AOT_SBR && (c->sbr = (int)bit1(&b)) == 1)
                    c->ext_sample_rate = get_sample_rate(&b, &c->ext_sampling_index);
                if (bits_left(&b) > 11 && bits(&b, 11) == 0x548)
                    c->ps = (int)bit1(&b);
                break;
            }
            bit1(&b);
        }
    }
    if (!c->sbr) c->ps = 0;                            /* PS requires SBR */
    const int channels = c->chan_config < 8 ? (c->chan_config == 7 ? 8 : c->chan_config) : 0;
    if ((c->ps == -1 && c->object_type != HEAAC_AOT_AAC_LC) || (channels & ~0x01))
        c->ps = 0;                                     /* implicit PS only in the HE-AACv2 profile */
    if (b.over) return HEAAC_PARSE_ERR_OVERREAD;
    return specific;
}

/* GASpecificConfig behind the AudioSpecificConfig (decode_ga_specific_config, aacdec.c:401-452): the
 * 960-sample frame length is refused as the reference refuses it; dependsOnCoreCoder / coreCoderDelay and
 * extensionFlag are read past.  channel_config 0 (a program config element follows) is outside this slice. */
int heaac_ga_specific_config(const HeaacAacConfig *c, const uint8_t *buf, int size, int bit_offset)
{
    if (!c || !buf || size <= 0 || bit_offset < 0) return HEAAC_PARSE_ERR_ARG;
    Bits b;
    bits_init(&b, buf, size);
    skip(&b, bit_offset);
    if (bit1(&b)) return HEAAC_PARSE_ERR_UNSUPPORTED;          /* frameLengthFlag: 960/120 MDCT window */
    if (bit1(&b)) skip(&b, 14);                                /* dependsOnCoreCoder: coreCoderDelay */
    bit1(&b);                                                  /* extensionFlag (no ER object types here) */
    if (c->chan_config == 0) return HEAAC_PARSE_ERR_UNSUPPORTED;
    if (b.over) return HEAAC_PARSE_ERR_OVERREAD;
    return 0;
}

int heaac_adts_parse_header(HeaacAdtsHeader *h, const uint8_t *buf, int size)
{
    if (!h || !buf || size < 7) return HEAAC_PARSE_ERR_ARG;
    Bits b;
    bits_init(&b, buf, size);
    if (bits(&b, 12) != 0xfff) return -1;
    bit1(&b);                                          /* id */
    bits(&b, 2);                                       /* layer */
    const int crc_abs = (int)bit1(&b);
    const int aot = (int)bits(&b, 2);
    const int sr = (int)bits(&b, 4);
    if (!k_rates[sr]) return -2;
    bit1(&b);                                          /* private_bit */
    const int ch = (int)bits(&b, 3);
    bits(&b, 4);                                       /* original/copy, home, copyright id bit + start */
    const int flen = (int)bits(&b, 13);
    if (flen < 7) return -3;
    bits(&b, 11);                                      /* adts_buffer_fullness */
    const int rdb = (int)bits(&b, 2);
    h->object_type = aot + 1;
    h->chan_config = ch;
    h->crc_absent = crc_abs;
    h->num_aac_frames = rdb + 1;
    h->sampling_index = sr;
    h->sample_rate = k_rates[sr];
    h->samples = (rdb + 1) * 1024;
    h->bit_rate = (int)((long long)flen * 8 * h->sample_rate / h->samples);
    h->frame_length = flen;
    return crc_abs ? 7 : 9;
}

/* ------------------------------------------------------------------------------------------ */
/* individual channel stream                                                                     */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int num_pulse, pos[4], amp[4];
} Pulse;

typedef struct {
    uint8_t window_sequence[2], use_kb_window[2];
} WinInfo;

/* How far the reference's element decoders had got when an access unit is refused: decoder state they have moved by
 * then stays moved (the window history of decode_ics_info, the noise generator of decode_spectrum_and_dequant, the
 * predictors of apply_prediction), although the unit gives no output.  `as_reference` is set where the refusal is
 * one the reference makes at the same bit of the unit; the checks of this parser alone (a read past the end that the
 * reference's unchecked reader would run through, the layouts it does not take) leave it clear. */
typedef struct {
    uint8_t ics[2];         /* per channel: 0 window history untouched, 1 moved on to this unit's, 2 cleared */
    uint8_t decoded[2];     /* decode_ics has returned 0 for the channel */
    uint8_t predicted[2];   /* apply_prediction has run on it */
    uint8_t as_reference;
    uint8_t ref_overread;   /* ... and it is one of the reference's own end-of-unit checks */
    int noise_stop[2];      /* channel not decoded: bands [0, noise_stop) have been through the spectrum loop */
} Progress;
#define REF_FAIL(pg, code) ((pg)->as_reference = 1, (code))
#define REF_OVERREAD(pg) ((pg)->as_reference = (pg)->ref_overread = 1, HEAAC_PARSE_ERR_OVERREAD)

/* decode_ics_info + decode_prediction (aacdec.c:622-742) */
static int read_ics_info(const HeaacAacConfig *cfg, Bits *b, HeaacToolsIcs *ics, HeaacPrediction *pred, WinInfo *w,
                         Progress *pg, int ch)
{
    const int si = cfg->sampling_index;
    /* every refusal in here is the reference's, and each of them clears the whole IndividualChannelStream
     * (memset, aacdec.c:650, 687, 692, 696, 705): the window history with it */
    pg->ics[ch] = 2;
    if (bit1(b)) return REF_FAIL(pg, HEAAC_PARSE_ERR_DATA);            /* reserved bit */
    w->window_sequence[1] = w->window_sequence[0];
    w->window_sequence[0] = (uint8_t)bits(b, 2);
    w->use_kb_window[1] = w->use_kb_window[0];
    w->use_kb_window[0] = (uint8_t)bit1(b);
    memset(ics, 0, sizeof(*ics));
    memset(pred, 0, sizeof(*pred));
    ics->num_window_groups = 1;
    ics->group_len[0] = 1;
    pred->pred_sfb_max = aac_pred_sfb_max[si];
    if (w->window_sequence[0] == HEAAC_EIGHT_SHORT_SEQUENCE) {
        ics->max_sfb = (uint8_t)bits(b, 4);
        for (int i = 0; i < 7; i++) {
            if (bit1(b)) {
                ics->group_len[ics->num_window_groups - 1]++;
            } else {
                ics->num_window_groups++;
                ics->group_len[ics->num_window_groups - 1] = 1;
            }
        }
        ics->num_windows = 8;
        ics->num_swb = aac_num_swb_128[si];
        ics->tns_max_bands = aac_tns_max_bands_128[si];
        memcpy(ics->swb_offset, aac_swb_offset_128 + aac_swb_first_128[si], (ics->num_swb + 1) * sizeof(uint16_t));
    } else {
        ics->max_sfb = (uint8_t)bits(b, 6);
        ics->num_windows = 1;
        ics->num_swb = aac_num_swb_1024[si];
        ics->tns_max_bands = aac_tns_max_bands_1024[si];
        memcpy(ics->swb_offset, aac_swb_offset_1024 + aac_swb_first_1024[si], (ics->num_swb + 1) * sizeof(uint16_t));
        pred->predictor_present = (uint8_t)bit1(b);
        if (pred->predictor_present) {
            if (cfg->object_type == HEAAC_AOT_AAC_MAIN) {
                if (bit1(b)) {
                    pred->predictor_reset_group = (uint8_t)bits(b, 5);
                    if (pred->predictor_reset_group == 0 || pred->predictor_reset_group > 30)
                        return REF_FAIL(pg, HEAAC_PARSE_ERR_DATA);
                }
                const int lim = ics->max_sfb < pred->pred_sfb_max ? ics->max_sfb : pred->pred_sfb_max;
                for (int sfb = 0; sfb < lim; sfb++) pred->prediction_used[sfb] = (uint8_t)bit1(b);
            } else if (cfg->object_type == HEAAC_AOT_AAC_LC) {
                return REF_FAIL(pg, HEAAC_PARSE_ERR_DATA);             /* prediction is not allowed in AAC-LC */
            } else {
                return REF_FAIL(pg, HEAAC_PARSE_ERR_UNSUPPORTED);      /* LTP */
            }
        }
    }
    if (ics->max_sfb > ics->num_swb) return REF_FAIL(pg, HEAAC_PARSE_ERR_DATA);
    pg->ics[ch] = 1;
    return HEAAC_PARSE_OK;
}

/* decode_band_types (:755-801) */
static int read_band_types(Bits *b, const HeaacToolsIcs *ics, int eight, uint8_t band_type[128], uint8_t run_end[128],
                           Progress *pg)
{
    const int nb = eight ? 3 : 5;
    int idx = 0;
    for (int g = 0; g < ics->num_window_groups; g++) {
        int k = 0;
        while (k < ics->max_sfb) {
            int sect_end = k, incr;
            const int bt = (int)bits(b, 4);
            if (bt == 12) return REF_FAIL(pg, HEAAC_PARSE_ERR_DATA);
            /* past the end the reference reads the zeros of its input padding, leaves the loop and fails its
             * get_bits_left() < 0 check (:778-781): the same refusal, taken here without the walk */
            while ((incr = (int)bits(b, nb)) == (1 << nb) - 1) {
                sect_end += incr;
                if (b->over) return REF_OVERREAD(pg);
            }
            sect_end += incr;
            if (b->over) return REF_OVERREAD(pg);
            if (sect_end > ics->max_sfb) return REF_FAIL(pg, HEAAC_PARSE_ERR_DATA);
            for (; k < sect_end; k++) {
                band_type[idx] = (uint8_t)bt;
                run_end[idx++] = (uint8_t)sect_end;
            }
        }
    }
    return HEAAC_PARSE_OK;
}

/* decode_scalefactors (:815-873) on the C path: sf_offset = 0 (+12 for eight short) */
static int read_scalefactors(Bits *b, const HeaacToolsIcs *ics, int eight, unsigned global_gain,
                             const uint8_t band_type[128], const uint8_t run_end[128], float sf[128], Progress *pg)
{
    const int sf_offset = eight ? 12 : 0;
    int offset[3] = { (int)global_gain, (int)global_gain - 90, 100 };
    int noise_flag = 1, idx = 0;
    for (int g = 0; g < ics->num_window_groups; g++) {
        for (int i = 0; i < ics->max_sfb;) {
            const int end = run_end[idx];
            const int bt = band_type[idx];
            if (bt == 0) {
                for (; i < end; i++, idx++) sf[idx] = 0.f;
            } else if (bt == HEAAC_INTENSITY_BT || bt == HEAAC_INTENSITY_BT2) {
                for (; i < end; i++, idx++) {
                    const int s = tree_read(&g_sf_tree, b);
                    if (s < 0) return HEAAC_PARSE_ERR_DATA;
                    offset[2] += s - 60;
                    if ((unsigned)offset[2] > 255U) return REF_FAIL(pg, HEAAC_PARSE_ERR_DATA);
                    sf[idx] = g_pow2sf[-offset[2] + 300];
                }
            } else if (bt == HEAAC_NOISE_BT) {
                for (; i < end; i++, idx++) {
                    if (noise_flag-- > 0) {
                        offset[1] += (int)bits(b, 9) - 256;
                    } else {
                        const int s = tree_read(&g_sf_tree, b);
                        if (s < 0) return HEAAC_PARSE_ERR_DATA;
                        offset[1] += s - 60;
                    }
                    if ((unsigned)offset[1] > 255U) return REF_FAIL(pg, HEAAC_PARSE_ERR_DATA);
                    sf[idx] = -g_pow2sf[offset[1] + sf_offset + 100];
                }
            } else {
                for (; i < end; i++, idx++) {
                    const int s = tree_read(&g_sf_tree, b);
                    if (s < 0) return HEAAC_PARSE_ERR_DATA;
                    offset[0] += s - 60;
                    if ((unsigned)offset[0] > 255U) return REF_FAIL(pg, HEAAC_PARSE_ERR_DATA);
                    sf[idx] = -g_pow2sf[offset[0] + sf_offset];
                }
            }
        }
    }
    return b->over ? HEAAC_PARSE_ERR_OVERREAD : HEAAC_PARSE_OK;
}

/* decode_pulses (:878-900) */
static int read_pulses(Bits *b, const HeaacToolsIcs *ics, Pulse *p, Progress *pg)
{
    p->num_pulse = (int)bits(b, 2) + 1;
    const int swb = (int)bits(b, 6);
    if (swb >= ics->num_swb) return REF_FAIL(pg, HEAAC_PARSE_ERR_DATA);
    p->pos[0] = ics->swb_offset[swb] + (int)bits(b, 5);
    if (p->pos[0] > 1023) return REF_FAIL(pg, HEAAC_PARSE_ERR_DATA);
    p->amp[0] = (int)bits(b, 4);
    for (int i = 1; i < p->num_pulse; i++) {
        p->pos[i] = (int)bits(b, 5) + p->pos[i - 1];
        if (p->pos[i] > 1023) return REF_FAIL(pg, HEAAC_PARSE_ERR_DATA);
        p->amp[i] = (int)bits(b, 4);
    }
    return HEAAC_PARSE_OK;
}

/* decode_tns (:907-945) */
static int read_tns(const HeaacAacConfig *cfg, Bits *b, const HeaacToolsIcs *ics, int eight, HeaacTns *tns, Progress *pg)
{
    const int max_order = eight ? 7 : cfg->object_type == HEAAC_AOT_AAC_MAIN ? 20 : 12;
    for (int w = 0; w < ics->num_windows; w++) {
        tns->n_filt[w] = (uint8_t)bits(b, 2 - eight);
        if (!tns->n_filt[w]) continue;
        const int coef_res = (int)bit1(b);
        for (int f = 0; f < tns->n_filt[w]; f++) {
            tns->length[w][f] = (uint8_t)bits(b, 6 - 2 * eight);
            tns->order[w][f] = (uint8_t)bits(b, 5 - 2 * eight);
            if (tns->order[w][f] > max_order) { tns->order[w][f] = 0; return REF_FAIL(pg, HEAAC_PARSE_ERR_DATA); }
            if (tns->order[w][f]) {
                tns->direction[w][f] = (uint8_t)bit1(b);
                const int compress = (int)bit1(b);
                const int len = coef_res + 3 - compress;
                const float *map = aac_tns_map[2 * compress + coef_res];
                for (int i = 0; i < tns->order[w][f]; i++)
                    tns->coef[w][f][i] = map[bits(b, len)];
            }
        }
    }
    return b->over ? HEAAC_PARSE_ERR_OVERREAD : HEAAC_PARSE_OK;
}

/* one quantised line of magnitude q (< 16 from the books, any from an escape) and sign -> float */
static inline float dequant(unsigned q, int negative, float sf)
{
    const float mag = q < 16 ? g_mag[q] : cbrtf((float)q) * (float)q;
    return (negative ? -mag : mag) * sf;
}

/* A pulse of amplitude `amp` on a line that has already been dequantised and scaled (:1222-1236): the reference
 * goes back to the quantised magnitude through float arithmetic -- x / sf, then x / x^(1/4) = |q| up to rounding --
 * adds the amplitude AWAY from zero (towards minus infinity on an empty line) and raises to 4/3 again as
 * cbrtf(|q|) * q.  All of it in float, in this order. */
static inline float pulse_add(float line, int amp, float sf)
{
    float q = -(float)amp;
    if (line != 0.0f) {
        const float x = line / sf;
        q = x / sqrtf(sqrtf(fabsf(x))) + (x > 0 ? (float)amp : -(float)amp);
    }
    return cbrtf(fabsf(q)) * q * sf;
}

/* decode_spectrum_and_dequant (:988-1245).  NOISE_BT bands are zeroed (the GPU fills them). */
static int read_spectrum(Bits *b, const HeaacToolsIcs *ics, const uint8_t band_type[128], const float sf[128],
                         int pulse_present, const Pulse *pulse, float coef[1024], Progress *pg, int ch)
{
    const int c = 1024 / ics->num_windows;
    const uint16_t *off = ics->swb_offset;
    float *base = coef;
    int idx = 0;
    for (int g = 0; g < ics->num_windows; g++)
        memset(coef + g * 128 + off[ics->max_sfb], 0, sizeof(float) * (c - off[ics->max_sfb]));
    for (int g = 0; g < ics->num_window_groups; g++) {
        const int g_len = ics->group_len[g];
        for (int i = 0; i < ics->max_sfb; i++, idx++) {
            const int bt = band_type[idx];
            float *cfo = coef + off[i];
            const int len = off[i + 1] - off[i];
            pg->noise_stop[ch] = idx;                  /* the noise bands in front of this one have drawn their numbers */
            if (bt == 0 || bt >= HEAAC_NOISE_BT) {
                for (int w = 0; w < g_len; w++) memset(cfo + 128 * w, 0, len * sizeof(float));
                continue;
            }
            const Tree *t = &g_spec_tree[bt - 1];
            const float s = sf[idx];
            for (int w = 0; w < g_len; w++) {
                float *cf = cfo + 128 * w;
                if (bt <= 4) {
                    /* quads: books 1, 2 signed (-1..1), books 3, 4 unsigned (0..2) + sign bits */
                    for (int k = 0; k < len; k += 4) {
                        const int code = tree_read(t, b);
                        if (code < 0) return HEAAC_PARSE_ERR_DATA;
                        int q[4] = { code / 27, code / 9 % 3, code / 3 % 3, code % 3 };
                        if (bt <= 2) {
                            for (int j = 0; j < 4; j++) {
                                const int v = q[j] - 1;
                                cf[k + j] = dequant((unsigned)abs(v), v < 0, s);
                            }
                        } else {
                            /* VMUL4S (:949-972) flips the scalefactor's sign by the sign bit at the head of the
                             * pending ones and moves on only behind a non-zero line: a zero line is multiplied by
                             * the scalefactor with the NEXT non-zero line's sign (none left: as it is) -- the sign of
                             * a zero, which no value downstream depends on, kept for the coefficients' bit pattern. */
                            int neg[4], pending = 0;
                            for (int j = 0; j < 4; j++) neg[j] = q[j] ? (int)bit1(b) : 0;
                            for (int j = 3; j >= 0; j--) {
                                if (q[j]) pending = neg[j];
                                else neg[j] = pending;
                            }
                            for (int j = 0; j < 4; j++) cf[k + j] = dequant((unsigned)q[j], neg[j], s);
                        }
                    }
                } else {
                    /* pairs: books 5, 6 signed (-4..4); 7, 8 (0..7), 9, 10 (0..12), 11 (0..16, 16 = escape) unsigned */
                    const int mod = bt <= 6 ? 9 : bt <= 8 ? 8 : bt <= 10 ? 13 : 17;
                    for (int k = 0; k < len; k += 2) {
                        const int code = tree_read(t, b);
                        if (code < 0) return HEAAC_PARSE_ERR_DATA;
                        int q[2] = { code / mod, code % mod };
                        if (bt <= 6) {
                            for (int j = 0; j < 2; j++) {
                                const int v = q[j] - 4;
                                cf[k + j] = dequant((unsigned)abs(v), v < 0, s);
                            }
                        } else {
                            int neg[2];
                            for (int j = 0; j < 2; j++) neg[j] = q[j] ? (int)bit1(b) : 0;
                            /* book 11 ORs the pending sign bit into a zero line as well (:1199-1201); books 7 ... 10
                             * (VMUL2S :935-947) do not */
                            if (bt == 11 && !q[0]) neg[0] = neg[1];
                            for (int j = 0; j < 2; j++) {
                                unsigned v = (unsigned)q[j];
                                if (bt == 11 && q[j] == 16) {
                                    /* escape_sequence: N ones, a zero, then N + 4 bits (:1174-1197) */
                                    int n = 0;
                                    while (bit1(b)) {
                                        if (++n > 8) return REF_FAIL(pg, HEAAC_PARSE_ERR_DATA);   /* "ESC overflow", :1187-1190 */
                                    }
                                    v = (1u << (n + 4)) + bits(b, n + 4);
                                }
                                cf[k + j] = dequant(v, neg[j], s);
                            }
                        }
                    }
                }
                if (b->over) return HEAAC_PARSE_ERR_OVERREAD;
            }
        }
        coef += g_len << 7;
    }
    if (pulse_present) {
        int band = 0;
        for (int i = 0; i < pulse->num_pulse; i++) {
            const int line = pulse->pos[i];
            while (off[band + 1] <= line) band++;
            /* no pulses into noise bands or bands without a scalefactor (:1227) */
            if (band_type[band] == HEAAC_NOISE_BT || sf[band] == 0.0f) continue;
            base[line] = pulse_add(base[line], pulse->amp[i], sf[band]);
        }
    }
    return HEAAC_PARSE_OK;
}

/* decode_ics (:1334-1388) without apply_prediction (a GPU stage) */
static int read_ics(const HeaacAacConfig *cfg, Bits *b, int common_window, HeaacToolsChannel *ch, WinInfo *w, float coef[1024],
                    Progress *pg, int c)
{
    Pulse pulse;
    pulse.num_pulse = 0;
    const unsigned global_gain = bits(b, 8);
    int r;
    if (!common_window && (r = read_ics_info(cfg, b, &ch->ics, &ch->pred, w, pg, c)) < 0)
        return r;
    const int eight = w->window_sequence[0] == HEAAC_EIGHT_SHORT_SEQUENCE;
    uint8_t run_end[128];
    memset(ch->band_type, 0, sizeof(ch->band_type));
    memset(ch->sf, 0, sizeof(ch->sf));
    memset(&ch->tns, 0, sizeof(ch->tns));
    if ((r = read_band_types(b, &ch->ics, eight, ch->band_type, run_end, pg)) < 0) return r;
    if ((r = read_scalefactors(b, &ch->ics, eight, global_gain, ch->band_type, run_end, ch->sf, pg)) < 0) return r;
    const int pulse_present = (int)bit1(b);
    if (pulse_present) {
        if (eight) return REF_FAIL(pg, HEAAC_PARSE_ERR_DATA);          /* pulse tool not allowed in eight short sequence */
        if ((r = read_pulses(b, &ch->ics, &pulse, pg)) < 0) return r;
    }
    ch->tns.present = (uint8_t)bit1(b);
    if (ch->tns.present && (r = read_tns(cfg, b, &ch->ics, eight, &ch->tns, pg)) < 0) return r;
    if (bit1(b)) return REF_FAIL(pg, HEAAC_PARSE_ERR_UNSUPPORTED);     /* gain control (SSR) */
    pg->noise_stop[c] = 0;
    if ((r = read_spectrum(b, &ch->ics, ch->band_type, ch->sf, pulse_present, &pulse, coef, pg, c)) < 0) return r;
    pg->decoded[c] = 1;
    /* apply_prediction inside decode_ics (:1381-1382) */
    if (cfg->object_type == HEAAC_AOT_AAC_MAIN && !common_window) pg->predicted[c] = 1;
    return HEAAC_PARSE_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* access unit                                                                                   */
/* ------------------------------------------------------------------------------------------ */
enum { TYPE_SCE, TYPE_CPE, TYPE_CCE, TYPE_LFE, TYPE_DSE, TYPE_PCE, TYPE_FIL, TYPE_END };
enum { EXT_DYNAMIC_RANGE = 0xb, EXT_SBR_DATA = 0xd, EXT_SBR_DATA_CRC = 0xe };

/* channel_pair_element behind its instance tag (decode_cpe, :1453-1492) without the spectral tools (GPU stages);
 * w[2] = the two channels' window history, coeffs [2][1024] */
static int read_cpe(const HeaacAacConfig *cfg, Bits *b, HeaacToolsFrame *tools, WinInfo *w, float *coeffs, Progress *pg)
{
    int r;
    const int common = (int)bit1(b);
    tools->common_window = (uint8_t)common;
    if (common) {
        if ((r = read_ics_info(cfg, b, &tools->ch[0].ics, &tools->ch[0].pred, &w[0], pg, 0)) < 0) return r;
        /* channel 1 takes channel 0's ics, keeping its own previous window shape (:1462-1464) */
        const uint8_t kb_prev1 = w[1].use_kb_window[0];
        w[1] = w[0];
        w[1].use_kb_window[1] = kb_prev1;
        pg->ics[1] = 1;
        tools->ch[1].ics = tools->ch[0].ics;
        tools->ch[1].pred = tools->ch[0].pred;
        tools->ms_present = (uint8_t)bits(b, 2);
        if (tools->ms_present == 3) return REF_FAIL(pg, HEAAC_PARSE_ERR_DATA);
        const int nb = tools->ch[0].ics.num_window_groups * tools->ch[0].ics.max_sfb;
        if (tools->ms_present == 1)
            for (int i = 0; i < nb; i++) tools->ms_mask[i] = (uint8_t)bit1(b);
        else if (tools->ms_present == 2)
            memset(tools->ms_mask, 1, nb);
    }
    if ((r = read_ics(cfg, b, common, &tools->ch[0], &w[0], coeffs, pg, 0)) < 0) return r;
    if ((r = read_ics(cfg, b, common, &tools->ch[1], &w[1], coeffs + 1024, pg, 1)) < 0) return r;
    /* apply_prediction at the end of decode_cpe (:1486-1489) */
    if (common && cfg->object_type == HEAAC_AOT_AAC_MAIN) pg->predicted[0] = pg->predicted[1] = 1;
    return HEAAC_PARSE_OK;
}

/* data_stream_element behind its tag (skip_data_stream_element, :602-620) */
static int skip_dse(Bits *b)
{
    const int align = (int)bit1(b);
    int count = (int)bits(b, 8);
    if (count == 255) count += (int)bits(b, 8);
    if (align) b->pos = (b->pos + 7) & ~7;
    if (bits_left(b) < 8 * count) return HEAAC_PARSE_ERR_OVERREAD;
    b->pos += 8 * count;
    return HEAAC_PARSE_OK;
}

/* program_config_element: read past (decode_pce, aacdec.c:303-357).  The reference turns it into a channel
 * layout (output_configure); this slice keeps the layout of the configuration. */
static int skip_pce(Bits *b)
{
    bits(b, 2);                                        /* object_type */
    bits(b, 4);                                        /* sampling_index */
    const int num_front = (int)bits(b, 4), num_side = (int)bits(b, 4), num_back = (int)bits(b, 4);
    const int num_lfe = (int)bits(b, 2), num_assoc = (int)bits(b, 3), num_cc = (int)bits(b, 4);
    if (bit1(b)) bits(b, 4);                           /* mono_mixdown_tag */
    if (bit1(b)) bits(b, 4);                           /* stereo_mixdown_tag */
    if (bit1(b)) bits(b, 3);                           /* mixdown_coeff_index, pseudo_surround */
    skip(b, 5 * (num_front + num_side + num_back));    /* is_cpe + tag per element */
    skip(b, 4 * num_lfe);
    skip(b, 4 * num_assoc);
    skip(b, 5 * num_cc);                               /* cc_element_is_ind_sw + tag */
    b->pos = (b->pos + 7) & ~7;
    const int comment = 8 * (int)bits(b, 8);
    if (bits_left(b) < comment) return HEAAC_PARSE_ERR_OVERREAD;
    skip(b, comment);
    return HEAAC_PARSE_OK;
}

/* decode_dynamic_range (:1596-1641) behind the payload's type nibble: nothing of it is used on this path, but it
 * says how long it is -- the one extension payload that does not take all that is left of its fill element. */
static int drc_bytes(Bits *b)
{
    int n = 1, bands = 1;
    if (bit1(b)) { skip(b, 8); n++; }                  /* pce_instance_tag, reserved */
    if (bit1(b)) {                                     /* excluded channels (decode_drc_channel_exclusions :1575-1587) */
        int num = 0;
        do { skip(b, 7); num += 7; } while (num < 64 - 7 && bit1(b));
        n += num / 7;
    }
    if (bit1(b)) {                                     /* band_incr, interpolation_scheme, band_top[] */
        bands += (int)bits(b, 4);
        skip(b, 4 + 8 * bands);
        n += 1 + bands;
    }
    if (bit1(b)) { skip(b, 8); n++; }                  /* prog_ref_level */
    skip(b, 8 * bands);                                /* dyn_rng_sgn, dyn_rng_ctl */
    return n + bands;
}

/* The body of a fill element of `cnt` bytes: extension payloads until they are used up (aac_decode_frame :2050-2060,
 * decode_extension_payload :1650-1690).  *sbr_bit = where an SBR payload starts (behind its type nibble; it takes
 * all that is left, :1044-1050), -1 for none. */
static void read_fil(Bits *b, int cnt, int *sbr_bit, int *sbr_bytes, int *sbr_crc)
{
    *sbr_bit = -1;
    while (cnt > 0) {
        const int type = (int)bits(b, 4);
        if (type == EXT_DYNAMIC_RANGE) {
            cnt -= drc_bytes(b);
            continue;
        }
        if (type == EXT_SBR_DATA || type == EXT_SBR_DATA_CRC) {
            *sbr_bit = b->pos;
            *sbr_bytes = cnt;
            *sbr_crc = type == EXT_SBR_DATA_CRC;
        }
        skip(b, 8 * cnt - 4);
        cnt = 0;
    }
}

/* coupling_channel_element (decode_cce, aacdec.c:1503-1570) */
typedef struct { int type, id, ch_select; } CceTarget;
/* A coupling element as transmitted: its target list and its gain lists in transmission order (:1538-1567).  Which
 * of the lists land on an output element is cce_resolve's. */
typedef struct {
    int num_coupled;
    CceTarget tg[8];
    float gl[16][120];
} CceLists;

/* One coupling gain.  The reference keeps the base in a `float scale` (aacdec.c:1508, :1528: 2^(1/8), 2^(1/4), 2^(1/2)
 * are ROUNDED to float before anything is raised to a power), calls the double pow() on it and rounds the result to
 * float (:1539, :1556; the sign is applied in double, which is exact). */
static float cce_gain(float base, int step, int negative)
{
    const double mag = pow((double)base, (double)-step);
    return (float)(negative ? -mag : mag);
}

/* The target list (:1511-1523).  Returns the number of gain lists that follow the channel stream: one per target,
 * two for a pair coupled with a gain list per channel. */
static int cce_read_targets(Bits *b, CceLists *ls)
{
    int lists = 0;
    ls->num_coupled = (int)bits(b, 3);
    for (int c = 0; c <= ls->num_coupled; c++) {
        CceTarget *t = &ls->tg[c];
        const int pair = (int)bit1(b);
        t->type = pair ? TYPE_CPE : TYPE_SCE;
        t->id = (int)bits(b, 4);
        t->ch_select = pair ? (int)bits(b, 2) : 2;
        lists += 1 + (t->ch_select == 3);
    }
    return lists;
}

/* How one gain list is coded. */
typedef struct {
    float base;            /* gain_element_scale as the reference's float */
    int sign_coded;        /* gain_element_sign: the low bit of an accumulated step is the sign */
    int common;            /* one gain for the whole list (always, for the first list: gain 1) */
    int first;             /* the first list carries no common gain: it starts from 1.0f */
} CceListCoding;

/* One gain list over the coupling channel's scalefactor bands (:1534-1566).  A list with a common gain holds that
 * value in every coded band; otherwise each coded band transmits a step that ACCUMULATES (`t = gain += t`), a zero
 * step repeating the value before it -- which, before the first non-zero step, is the list's starting value. */
static int cce_read_gain_list(Bits *b, const CceListCoding *k, const HeaacToolsChannel *ch, int after_imdct,
                              float out[120])
{
    int acc = 0;
    float cur = 1.0f;
    if (!k->first) {
        if (k->common) {
            const int sym = tree_read(&g_sf_tree, b);
            if (sym < 0) return HEAAC_PARSE_ERR_DATA;
            acc = sym - 60;
        }
        cur = cce_gain(k->base, acc, 0);               /* unsigned whatever gain_element_sign says (:1539) */
    }
    if (after_imdct) {
        out[0] = cur;
        return HEAAC_PARSE_OK;
    }
    const int n_bands = ch->ics.num_window_groups * ch->ics.max_sfb;
    if (n_bands > 120) return HEAAC_PARSE_ERR_DATA;
    for (int band = 0; band < n_bands; band++) {
        if (ch->band_type[band] == 0) continue;        /* ZERO_BT: no gain, no bits */
        if (!k->common) {
            const int sym = tree_read(&g_sf_tree, b);
            if (sym < 0) return HEAAC_PARSE_ERR_DATA;
            if (sym != 60) {
                acc += sym - 60;
                /* arithmetic shift of the accumulated step, the low bit being the sign (:1552-1555) */
                cur = k->sign_coded ? cce_gain(k->base, acc >> 1, acc & 1) : cce_gain(k->base, acc, 0);
            }
        }
        out[band] = cur;
    }
    return HEAAC_PARSE_OK;
}

static int read_cce(const HeaacAacConfig *cfg, Bits *b, int elem_id, HeaacCceFrame *out, CceLists *ls,
                    HeaacToolsChannel *ch, WinInfo *w, float coef[1024])
{
    memset(out, 0, sizeof(*out));
    out->present = 1;
    out->elem_id = (uint8_t)elem_id;
    const int independent = (int)bit1(b);              /* ind_sw_cce_flag */
    const int n_lists = cce_read_targets(b, ls);
    const int after_tns = (int)bit1(b);                /* cc_domain; read either way (:1524) */
    const int point = independent ? HEAAC_CC_AFTER_IMDCT : after_tns;
    out->coupling_point = (uint8_t)point;
    CceListCoding k;
    k.sign_coded = (int)bit1(b);
    k.base = (float)pow(2., pow(2., (int)bits(b, 2) - 3));
    Progress pg;                                       /* a refusal inside a coupling element is not followed up */
    memset(&pg, 0, sizeof(pg));
    int r = read_ics(cfg, b, 0, ch, w, coef, &pg, 0);
    if (r < 0) return r;
    out->ics = ch->ics;
    memcpy(out->band_type, ch->band_type, sizeof(out->band_type));

    memset(ls->gl, 0, sizeof(ls->gl));
    for (int c = 0; c < n_lists; c++) {
        k.first = c == 0;
        k.common = k.first || point == HEAAC_CC_AFTER_IMDCT || bit1(b);
        if ((r = cce_read_gain_list(b, &k, ch, point == HEAAC_CC_AFTER_IMDCT, ls->gl[c])) < 0) return r;
    }
    return HEAAC_PARSE_OK;
}

/* Which lists land on the output element (target_type, target_id), and on which of its channels: the index walk of
 * apply_channel_coupling (:1870-1898) -- every entry of the target list consumes one gain list, or two for a pair
 * coupled with separate gains (ch_select 3), whether or not it names this element. */
static int cce_resolve(const CceLists *ls, int target_type, int target_id, HeaacCceFrame *out)
{
    int index = 0, n_links = 0;
    for (int c = 0; c <= ls->num_coupled; c++) {
        const CceTarget *tg = &ls->tg[c];
        if (tg->type == target_type && tg->id == target_id) {
            int use[2], nuse = 0, chn[2];
            if (tg->ch_select != 1) {
                use[nuse] = index; chn[nuse++] = 0;
                if (tg->ch_select != 0) index++;
            }
            if (tg->ch_select != 2) { use[nuse] = index++; chn[nuse++] = 1; }
            for (int k = 0; k < nuse; k++) {
                if (n_links >= HEAAC_MAX_CCE_LINKS) return HEAAC_PARSE_ERR_UNSUPPORTED;
                out->link[n_links].target_ch = (uint8_t)chn[k];
                memcpy(out->link[n_links].gain, ls->gl[use[k]], sizeof(out->link[n_links].gain));
                n_links++;
            }
        } else {
            index += 1 + (tg->ch_select == 3);
        }
    }
    out->n_links = (uint8_t)n_links;
    return HEAAC_PARSE_OK;
}

int heaac_aac_parse_frame(const HeaacAacConfig *cfg, HeaacAacStream *st,
                          const uint8_t *au, int size,
                          float *coeffs, HeaacIcs *ics, HeaacToolsFrame *tools,
                          HeaacAacFrameInfo *info)
{
    return heaac_aac_parse_frame_ex(cfg, st, au, size, 2, coeffs, ics, tools, NULL, info);
}

/* get_che (aacdec.c:113-177) for a stream of channel configuration 1 or 2: the configuration's one element -- an SCE
 * for 1, a CPE for 2 -- is mapped to the instance tag it is first met with; anything else, a second element of the
 * unit (its tag counts as seen, the next one is not mapped) and the same element under another tag in a later unit
 * find no element allocated and fail the unit there (:2011-2015), with nothing of their own read. */
static int output_element_allowed(const HeaacAacConfig *cfg, HeaacAacStream *st, int type, int tag, int have_one, Progress *pg)
{
    const int one_element = cfg->chan_config == 1 || cfg->chan_config == 2;
    if (have_one) return one_element ? REF_FAIL(pg, HEAAC_PARSE_ERR_UNSUPPORTED) : HEAAC_PARSE_ERR_UNSUPPORTED;
    if (one_element) {
        if ((type == TYPE_CPE) != (cfg->chan_config == 2)) return REF_FAIL(pg, HEAAC_PARSE_ERR_DATA);
        if (st->mapped_tag && st->mapped_tag != tag + 1) return REF_FAIL(pg, HEAAC_PARSE_ERR_DATA);
        st->mapped_tag = (uint8_t)(tag + 1);           /* (kept whatever becomes of the unit, as tag_che_map is) */
    }
    return HEAAC_PARSE_OK;
}

/* The walk over one access unit of a one-element stream; `b`, `w`, `pg`, `n_cce_seen` are the caller's so that it
 * can tell, after a refusal, how far the walk had got. */
static int frame_walk(const HeaacAacConfig *cfg, HeaacAacStream *st, const uint8_t *au, int size, int coeff_channels,
                      float *coeffs, HeaacIcs *ics, HeaacToolsFrame *tools, const HeaacCceOut *cce,
                      HeaacAacFrameInfo *info, Bits *b, WinInfo w[2], Progress *pg, int *n_cce_seen)
{
    bits_init(b, au, size);
    if (peek(b, 12) == 0xfff) {
        /* an ADTS header in front of the raw data block (aacdec.c:1988-1997) */
        HeaacAdtsHeader h;
        const int hs = heaac_adts_parse_header(&h, au, size);
        if (hs < 0) return HEAAC_PARSE_ERR_DATA;
        /* parse_adts_frame_header (aacdec.c:1935-1971) takes rate and object type from every header and refuses
         * more than one raw data block per frame.  `cfg` is the caller's (read-only, shared by a batch): a header
         * that contradicts it would be dequantised against the wrong band tables, so it is refused instead. */
        if (h.num_aac_frames != 1) return HEAAC_PARSE_ERR_UNSUPPORTED;
        if (h.sampling_index != cfg->sampling_index || h.object_type != cfg->object_type) return HEAAC_PARSE_ERR_DATA;
        b->pos = hs * 8;
    }
    WinInfo wc[HEAAC_MAX_CCE];
    for (int c = 0; c < 2; c++) {
        w[c].window_sequence[0] = st->window_sequence[c];
        w[c].use_kb_window[0] = st->use_kb_window[c];
        w[c].window_sequence[1] = w[c].use_kb_window[1] = 0;
    }
    HeaacAacFrameInfo fi = { 0, 0, -1, 0, 0, 0, 0, 0, 0 };
    int last_che = 0, prev_type = TYPE_END;            /* 1 + type of the channel element last seen; the element in front */
    /* The coupling elements name their targets by (type, tag): the output element of this slice is the one SCE /
     * CPE of the configuration (set_default_channel_config: tag 0), known before the walk starts. */
    const int target_type = cfg->chan_config == 2 ? TYPE_CPE : TYPE_SCE;
    int cce_tag[HEAAC_MAX_CCE], n_cce = 0;
    if (cce) memset(cce->cce, 0, HEAAC_MAX_CCE * sizeof(HeaacCceFrame));
    int elem, r;
    while ((elem = (int)bits(b, 3)) != TYPE_END) {
        int elem_id = (int)bits(b, 4);
        switch (elem) {
        case TYPE_SCE:
            if ((r = output_element_allowed(cfg, st, TYPE_SCE, elem_id, fi.channels, pg)) < 0) return r;
            if ((r = read_ics(cfg, b, 0, &tools->ch[0], &w[0], coeffs, pg, 0)) < 0) return r;
            fi.channels = 1;
            fi.elem_id = elem_id;
            break;
        case TYPE_CPE: {
            if ((r = output_element_allowed(cfg, st, TYPE_CPE, elem_id, fi.channels, pg)) < 0) return r;
            if (coeff_channels < 2) return HEAAC_PARSE_ERR_ARG;
            if ((r = read_cpe(cfg, b, tools, w, coeffs, pg)) < 0) return r;
            fi.channels = 2;
            fi.elem_id = elem_id;
            break;
        }
        case TYPE_CCE: {
            if (!cce || n_cce >= HEAAC_MAX_CCE) return HEAAC_PARSE_ERR_UNSUPPORTED;
            for (int k = 0; k < n_cce; k++)
                if (cce_tag[k] == elem_id) return HEAAC_PARSE_ERR_DATA;           /* the same element twice */
            /* slots in ascending tag order: a smaller tag arriving later moves the earlier element up */
            int slot = n_cce;
            while (slot > 0 && cce_tag[slot - 1] > elem_id) {
                cce->cce[slot] = cce->cce[slot - 1];
                cce->tools[slot] = cce->tools[slot - 1];
                memcpy(cce->coeffs + slot * 1024, cce->coeffs + (slot - 1) * 1024, 4096);
                wc[slot] = wc[slot - 1];
                cce_tag[slot] = cce_tag[slot - 1];
                slot--;
            }
            cce_tag[slot] = elem_id;
            n_cce++;
            *n_cce_seen = n_cce;
            /* the coupling channel's window history: by instance tag, as the reference keeps it (che[TYPE_CCE][tag]) --
             * slots move with the tags an access unit happens to carry and the order they arrive in */
            wc[slot].window_sequence[0] = st->cce_window_sequence[elem_id];
            wc[slot].use_kb_window[0] = st->cce_use_kb_window[elem_id];
            wc[slot].window_sequence[1] = wc[slot].use_kb_window[1] = 0;
            memset(&cce->tools[slot], 0, sizeof(HeaacToolsFrame));
            CceLists ls;
            r = read_cce(cfg, b, elem_id, &cce->cce[slot], &ls, &cce->tools[slot].ch[0], &wc[slot],
                         cce->coeffs + slot * 1024);
            if (r < 0) return r;
            if ((r = cce_resolve(&ls, target_type, 0, &cce->cce[slot])) < 0) return r;
            cce->cce[slot].behind_target = cce->cce[slot].outputs_before = (uint8_t)(fi.channels != 0);
            cce->cce[slot].seq = (uint8_t)(n_cce - 1);
            break;
        }
        case TYPE_LFE:
            /* no LFE in a one- or two-channel layout: get_che finds no element for it ("channel element %d.%d is not
             * allocated", :2011-2015) */
            return REF_FAIL(pg, HEAAC_PARSE_ERR_UNSUPPORTED);
        case TYPE_DSE: {
            if ((r = skip_dse(b)) < 0) return r;
            break;
        }
        case TYPE_PCE:
            if ((r = skip_pce(b)) < 0) return r;
            break;
        case TYPE_FIL: {
            if (elem_id == 15) elem_id += (int)bits(b, 8) - 1;
            if (bits_left(b) < 8 * elem_id) return REF_OVERREAD(pg);                  /* :2053-2056 */
            /* an SBR payload is located here and parsed by sbr_parse.c.  decode_extension_payload hands it to the
             * channel element last seen, together with the type of the element directly in front (:2059) */
            int at, bytes = 0, crc = 0;
            read_fil(b, elem_id, &at, &bytes, &crc);
            if (at >= 0) {
                if (!last_che) return HEAAC_PARSE_ERR_DATA;            /* "SBR was found before the first channel element" */
                if (last_che == TYPE_CCE + 1) return HEAAC_PARSE_ERR_UNSUPPORTED;   /* the coupling element's own SBR */
                if (fi.sbr_payload_bit >= 0) return HEAAC_PARSE_ERR_UNSUPPORTED;    /* a second payload for the element */
                fi.sbr_payload_bit = at;
                fi.sbr_payload_bytes = bytes;
                fi.sbr_crc = crc;
                fi.sbr_misplaced = prev_type != TYPE_SCE && prev_type != TYPE_CPE;
            }
            break;
        }
        default:
            return HEAAC_PARSE_ERR_UNSUPPORTED;
        }
        if (elem < TYPE_DSE) last_che = elem + 1;
        prev_type = elem;
        if (b->over) return HEAAC_PARSE_ERR_OVERREAD;
        if (bits_left(b) < 3) return REF_OVERREAD(pg);                             /* :2072-2075 */
    }
    if (!fi.channels) return HEAAC_PARSE_ERR_DATA;
    if (fi.elem_id != 0 && n_cce) {
        /* the links were resolved against tag 0 (the default layout's): an output element with another tag is only
         * reachable through a program config element */
        return HEAAC_PARSE_ERR_UNSUPPORTED;
    }
    for (int c = 0; c < fi.channels; c++) {
        ics[c].window_sequence[0] = w[c].window_sequence[0];
        ics[c].window_sequence[1] = w[c].window_sequence[1];
        ics[c].use_kb_window[0] = w[c].use_kb_window[0];
        ics[c].use_kb_window[1] = w[c].use_kb_window[1];
        st->window_sequence[c] = w[c].window_sequence[0];
        st->use_kb_window[c] = w[c].use_kb_window[0];
    }
    for (int k = 0; k < n_cce; k++) {
        cce->ics[k].window_sequence[0] = wc[k].window_sequence[0];
        cce->ics[k].window_sequence[1] = wc[k].window_sequence[1];
        cce->ics[k].use_kb_window[0] = wc[k].use_kb_window[0];
        cce->ics[k].use_kb_window[1] = wc[k].use_kb_window[1];
        st->cce_window_sequence[cce_tag[k]] = wc[k].window_sequence[0];
        st->cce_use_kb_window[cce_tag[k]] = wc[k].use_kb_window[0];
    }
    fi.n_cce = n_cce;
    fi.bits_consumed = b->pos;
    if (info) *info = fi;
    return HEAAC_PARSE_OK;
}

/* A channel record that only draws `draws` numbers from the noise generator (bands of at most 96 lines, the widest
 * the band tables have) and leaves the predictors alone: one long window, so that no reset applies. */
static void noise_only_channel(HeaacToolsChannel *ch, float *coef, int draws)
{
    memset(ch, 0, sizeof(*ch));
    if (coef) memset(coef, 0, 1024 * sizeof(float));
    ch->ics.num_windows = ch->ics.num_window_groups = ch->ics.group_len[0] = 1;
    int nb = 0, at = 0;
    while (draws > 0) {
        const int len = draws < 96 ? draws : 96;
        ch->ics.swb_offset[nb] = (uint16_t)at;
        ch->band_type[nb] = HEAAC_NOISE_BT;
        ch->sf[nb] = 1.0f;
        at += len;
        draws -= len;
        nb++;
    }
    ch->ics.swb_offset[nb] = (uint16_t)at;
    ch->ics.max_sfb = ch->ics.num_swb = (uint8_t)nb;
}

/* The noise bands among the first `stop` bands of a channel, in lines (= numbers drawn, :1016-1029) */
static int noise_draws(const HeaacToolsChannel *ch, int stop)
{
    int idx = 0, draws = 0;
    for (int g = 0; g < ch->ics.num_window_groups; g++)
        for (int i = 0; i < ch->ics.max_sfb; i++, idx++)
            if (idx < stop && ch->band_type[idx] == HEAAC_NOISE_BT)
                draws += ch->ics.group_len[g] * (ch->ics.swb_offset[i + 1] - ch->ics.swb_offset[i]);
    return draws;
}

/* A refused access unit gives no samples, but what the reference's element decoders did before they gave up is not
 * undone (aac_decode_frame returns from the middle of its element loop, :2069-2070): the window history that
 * decode_ics_info moved on or cleared, the numbers decode_spectrum_and_dequant drew for the noise bands it had passed,
 * the predictors apply_prediction stepped for a channel it completed.  Where the refusal is the reference's own,
 * `st` takes the same history here, and `tools` / `coeffs` are rewritten into records that make the spectral tools
 * draw and predict exactly that much (HEAAC_REFUSED_RUN_TOOLS); their coefficients are of no further use. */
static void unit_refused(const HeaacAacConfig *cfg, HeaacAacStream *st, const Bits *b, const WinInfo w[2],
                         const Progress *pg, int n_cce, int coeff_channels, float *coeffs, HeaacToolsFrame *tools,
                         HeaacAacFrameInfo *info)
{
    HeaacAacFrameInfo fi = { 0, 0, -1, 0, 0, 0, 0, 0, 0 };
    if (pg->as_reference && (!b->over || pg->ref_overread) && !n_cce) {
        fi.refused = HEAAC_REFUSED_AS_REFERENCE;
        for (int c = 0; c < 2; c++) {
            if (pg->ics[c] == 1) {
                st->window_sequence[c] = w[c].window_sequence[0];
                st->use_kb_window[c] = w[c].use_kb_window[0];
            } else if (pg->ics[c] == 2) {
                st->window_sequence[c] = st->use_kb_window[c] = 0;
            }
        }
        const int main_profile = cfg->object_type == HEAAC_AOT_AAC_MAIN;
        const int complete = pg->decoded[0] && (pg->ics[1] == 0 || pg->decoded[1]) &&
                             (!main_profile || (pg->predicted[0] && (pg->ics[1] == 0 || pg->predicted[1])));
        int work = 0;
        if (complete) {
            /* the element was decoded to its end (the refusal came behind it): its records stand as they are */
            work = main_profile || noise_draws(&tools->ch[0], 128) || (pg->ics[1] && noise_draws(&tools->ch[1], 128));
        } else {
            tools->common_window = tools->ms_present = 0;
            memset(tools->ms_mask, 0, sizeof(tools->ms_mask));
            for (int c = 0; c < 2; c++) {
                HeaacToolsChannel *ch = &tools->ch[c];
                float *coef = c < coeff_channels ? coeffs + c * 1024 : NULL;
                if (pg->decoded[c] && (!main_profile || pg->predicted[c])) {
                    work |= main_profile || noise_draws(ch, 128);
                    memset(&ch->tns, 0, sizeof(ch->tns));
                } else {
                    /* a channel that stopped inside its spectrum, or one whose prediction was still to come at the
                     * end of the pair: only its noise bands have left a trace */
                    const int draws = pg->ics[c] == 1 ? noise_draws(ch, pg->decoded[c] ? 128 : pg->noise_stop[c]) : 0;
                    noise_only_channel(ch, coef, draws);
                    work |= draws;
                }
            }
        }
        if (work) fi.refused |= HEAAC_REFUSED_RUN_TOOLS;
    }
    if (info) *info = fi;
}

int heaac_aac_parse_frame_ex(const HeaacAacConfig *cfg, HeaacAacStream *st,
                             const uint8_t *au, int size, int coeff_channels,
                             float *coeffs, HeaacIcs *ics, HeaacToolsFrame *tools,
                             const HeaacCceOut *cce, HeaacAacFrameInfo *info)
{
    if (!cfg || !st || !au || size <= 0 || !coeffs || !ics || !tools ||
        cfg->sampling_index < 0 || cfg->sampling_index > 12 || coeff_channels < 1 || coeff_channels > 2 ||
        (cce && (!cce->cce || !cce->coeffs || !cce->ics || !cce->tools)))
        return HEAAC_PARSE_ERR_ARG;
    pthread_once(&g_once, tables_init);
    if (g_tables_bad) return HEAAC_PARSE_ERR_ARG;
    Bits b;
    WinInfo w[2];
    Progress pg;
    int n_cce = 0;
    memset(&pg, 0, sizeof(pg));
    memset(w, 0, sizeof(w));
    memset(tools, 0, sizeof(*tools));
    const int r = frame_walk(cfg, st, au, size, coeff_channels, coeffs, ics, tools, cce, info, &b, w, &pg, &n_cce);
    if (r < 0) unit_refused(cfg, st, &b, w, &pg, n_cce, coeff_channels, coeffs, tools, info);
    return r;
}

/* ------------------------------------------------------------------------------------------ */
/* channel layouts: several output elements per access unit                                      */
/* ------------------------------------------------------------------------------------------ */
typedef struct { uint8_t type, id; } ElemRef;
/* The channel configurations 1..7 twice: the order the elements' channels leave the decoder
 * (aac_channel_layout_map, aacdectab.h:74-82) and the order the elements arrive in an access unit, which is what
 * get_che maps by (:138-181: the n-th output element of the stream must have the type standing here; where a
 * 5.1 / 7.1 layout has its LFE an SCE is taken too).  Mask = aac_channel_layout[] (aacdectab.h:84-93). */
static const struct ChanConfig {
    int n;
    ElemRef out[5], arrive[5];
    int64_t mask;
} k_chan_config[8] = {
    { 0, {{0, 0}}, {{0, 0}}, 0 },
    { 1, {{TYPE_SCE, 0}},                                                        {{TYPE_SCE, 0}}, 0x4 },
    { 1, {{TYPE_CPE, 0}},                                                        {{TYPE_CPE, 0}}, 0x3 },
    { 2, {{TYPE_CPE, 0}, {TYPE_SCE, 0}},                                         {{TYPE_SCE, 0}, {TYPE_CPE, 0}}, 0x7 },
    { 3, {{TYPE_CPE, 0}, {TYPE_SCE, 0}, {TYPE_SCE, 1}},                          {{TYPE_SCE, 0}, {TYPE_CPE, 0}, {TYPE_SCE, 1}}, 0x107 },
    { 3, {{TYPE_CPE, 0}, {TYPE_SCE, 0}, {TYPE_CPE, 1}},                          {{TYPE_SCE, 0}, {TYPE_CPE, 0}, {TYPE_CPE, 1}}, 0x37 },
    { 4, {{TYPE_CPE, 0}, {TYPE_SCE, 0}, {TYPE_LFE, 0}, {TYPE_CPE, 1}},           {{TYPE_SCE, 0}, {TYPE_CPE, 0}, {TYPE_CPE, 1}, {TYPE_LFE, 0}}, 0x3f },
    { 5, {{TYPE_CPE, 0}, {TYPE_SCE, 0}, {TYPE_LFE, 0}, {TYPE_CPE, 2}, {TYPE_CPE, 1}},
         {{TYPE_SCE, 0}, {TYPE_CPE, 0}, {TYPE_CPE, 1}, {TYPE_CPE, 2}, {TYPE_LFE, 0}}, 0xff },
};

static int layout_add(HeaacAacLayout *l, int type, int id)
{
    const int nch = type == TYPE_CPE ? 2 : 1;
    if (l->n_elements >= HEAAC_MAX_ELEMENTS || l->channels + nch > HEAAC_MAX_LAYOUT_CHANNELS) return HEAAC_PARSE_ERR_UNSUPPORTED;
    HeaacAacElementSlot *e = &l->elem[l->n_elements];
    e->type = (uint8_t)type; e->id = (uint8_t)id; e->channels = (uint8_t)nch; e->first_channel = (uint8_t)l->channels;
    l->slot_of[type][id] = (int8_t)(l->n_elements + 1);
    l->n_elements++;
    l->channels += nch;
    return 0;
}

int heaac_aac_layout_default(HeaacAacLayout *l, int chan_config)
{
    if (!l) return HEAAC_PARSE_ERR_ARG;
    memset(l, 0, sizeof(*l));
    if (chan_config < 1 || chan_config > 7) return HEAAC_PARSE_ERR_DATA;   /* "invalid default channel configuration" */
    const struct ChanConfig *c = &k_chan_config[chan_config];
    l->chan_config = chan_config;
    for (int i = 0; i < c->n; i++) layout_add(l, c->out[i].type, c->out[i].id);
    l->channel_layout = c->mask;
    return 0;
}

int heaac_aac_layout_from_pce(HeaacAacLayout *l, const uint8_t *buf, int size, int bit_offset, int *bits_used)
{
    if (!l || !buf || size <= 0 || bit_offset < 0) return HEAAC_PARSE_ERR_ARG;
    Bits b;
    bits_init(&b, buf, size);
    skip(&b, bit_offset);
    uint8_t have[4][16];
    memset(have, 0, sizeof(have));
    bits(&b, 2);                                       /* object_type */
    bits(&b, 4);                                       /* sampling_index (a mismatch with the configuration only warns) */
    const int num[3] = { (int)bits(&b, 4), (int)bits(&b, 4), (int)bits(&b, 4) };   /* front, side, back */
    const int num_lfe = (int)bits(&b, 2), num_assoc = (int)bits(&b, 3), num_cc = (int)bits(&b, 4);
    if (bit1(&b)) bits(&b, 4);                         /* mono_mixdown_tag */
    if (bit1(&b)) bits(&b, 4);                         /* stereo_mixdown_tag */
    if (bit1(&b)) bits(&b, 3);                         /* mixdown_coeff_index, pseudo_surround */
    for (int g = 0; g < 3; g++)
        for (int i = 0; i < num[g]; i++) {
            const int pair = (int)bit1(&b);
            have[pair ? TYPE_CPE : TYPE_SCE][bits(&b, 4)] = 1;
        }
    for (int i = 0; i < num_lfe; i++) have[TYPE_LFE][bits(&b, 4)] = 1;
    skip(&b, 4 * num_assoc);
    for (int i = 0; i < num_cc; i++) {
        bit1(&b);                                      /* cc_element_is_ind_sw: the element says so itself */
        have[TYPE_CCE][bits(&b, 4)] = 1;
    }
    b.pos = (b.pos + 7) & ~7;
    const int comment = 8 * (int)bits(&b, 8);
    if (b.over || bits_left(&b) < comment) return HEAAC_PARSE_ERR_OVERREAD;
    skip(&b, comment);
    memset(l, 0, sizeof(*l));
    /* output_configure without a channel configuration (:253-268): ids ascending, per id SCE, CPE, (CCE,) LFE */
    for (int id = 0; id < 16; id++) {
        static const int order[3] = { TYPE_SCE, TYPE_CPE, TYPE_LFE };
        for (int t = 0; t < 3; t++)
            if (have[order[t]][id] && layout_add(l, order[t], id) < 0) return HEAAC_PARSE_ERR_UNSUPPORTED;
    }
    /* the coupling elements the program names (che_configure allocates no others, :198-212): not output elements;
     * their slots, in ascending tag order, are the order apply_channel_coupling walks them in (:1876) */
    for (int id = 0, k = 0; id < 16; id++)
        if (have[TYPE_CCE][id]) l->slot_of[TYPE_CCE][id] = (int8_t)++k;
    memcpy(l->tag_map, l->slot_of, sizeof(l->tag_map));      /* tag_che_map = che: elements are found by their tag */
    l->tags_mapped = 4 * 16;
    if (bits_used) *bits_used = b.pos - bit_offset;
    return 0;
}

int heaac_aac_layout_from_au(HeaacAacLayout *l, const uint8_t *au, int size)
{
    if (!l || !au || size <= 0) return HEAAC_PARSE_ERR_ARG;
    Bits b;
    bits_init(&b, au, size);
    if (peek(&b, 12) == 0xfff) {
        HeaacAdtsHeader h;
        const int hs = heaac_adts_parse_header(&h, au, size);
        if (hs < 0) return HEAAC_PARSE_ERR_DATA;
        b.pos = hs * 8;
    }
    /* aac_decode_frame's element loop (:1999-2075) as far as the first program config element: with nothing allocated
     * yet only data stream and fill elements can stand in front of it */
    int type, r;
    while ((type = (int)bits(&b, 3)) != TYPE_END) {
        int tag = (int)bits(&b, 4);
        if (type == TYPE_PCE) {
            if (b.over) return HEAAC_PARSE_ERR_OVERREAD;
            return heaac_aac_layout_from_pce(l, au, size, b.pos, NULL);
        }
        if (type == TYPE_DSE) {
            if ((r = skip_dse(&b)) < 0) return r;
        } else if (type == TYPE_FIL) {
            if (tag == 15) tag += (int)bits(&b, 8) - 1;
            if (bits_left(&b) < 8 * tag) return HEAAC_PARSE_ERR_OVERREAD;
            int at, bytes, crc;
            read_fil(&b, tag, &at, &bytes, &crc);
            if (at >= 0) return HEAAC_PARSE_ERR_DATA;                  /* "SBR was found before the first channel element" */
        } else {
            return HEAAC_PARSE_ERR_DATA;                               /* "channel element %d.%d is not allocated" */
        }
        if (b.over || bits_left(&b) < 3) return HEAAC_PARSE_ERR_OVERREAD;
    }
    return HEAAC_PARSE_ERR_DATA;                                       /* no program: nothing this stream could decode */
}

int heaac_asc_layout(HeaacAacConfig *c, HeaacAacLayout *l, const uint8_t *buf, int size)
{
    if (!c || !l) return HEAAC_PARSE_ERR_ARG;
    const int specific = heaac_asc_parse(c, buf, size);
    if (specific < 0) return specific;
    /* decode_ga_specific_config (:401-452) */
    Bits b;
    bits_init(&b, buf, size);
    skip(&b, specific);
    if (bit1(&b)) return HEAAC_PARSE_ERR_UNSUPPORTED;          /* frameLengthFlag: 960/120 MDCT window */
    if (bit1(&b)) skip(&b, 14);                                /* dependsOnCoreCoder: coreCoderDelay */
    bit1(&b);                                                  /* extensionFlag (no ER object types here) */
    if (b.over) return HEAAC_PARSE_ERR_OVERREAD;
    if (c->chan_config) return heaac_aac_layout_default(l, c->chan_config);
    skip(&b, 4);                                               /* element_instance_tag of the program config element */
    if (b.over) return HEAAC_PARSE_ERR_OVERREAD;
    return heaac_aac_layout_from_pce(l, buf, size, b.pos, NULL);
}

/* get_che (:113-183): the slot of the layout a bitstream element (type, tag) lands in, or -1 */
static int layout_find(HeaacAacLayout *l, uint8_t seen[4][16], int type, int *tag_io)
{
    int tag = *tag_io;
    /* "Some buggy encoders appear to set all elem_ids to zero": a tag met twice in one access unit moves up */
    while (tag < 16 && seen[type][tag]) tag++;
    if (tag == 16) return -1;
    seen[type][tag] = 1;
    *tag_io = tag;
    if (l->tag_map[type][tag]) return l->tag_map[type][tag] - 1;
    if (l->chan_config < 1 || l->chan_config > 7) return -1;
    const struct ChanConfig *c = &k_chan_config[l->chan_config];
    if (l->tags_mapped >= c->n) return -1;
    const ElemRef want = c->arrive[l->tags_mapped];
    if (type != want.type && !(want.type == TYPE_LFE && type == TYPE_SCE)) return -1;
    const int slot = l->slot_of[want.type][want.id] - 1;
    l->tag_map[type][tag] = (int8_t)(slot + 1);
    l->tags_mapped++;
    return slot;
}

int heaac_aac_parse_frame_layout(const HeaacAacConfig *cfg, HeaacAacLayout *layout, HeaacAacStream *st,
                                 const uint8_t *au, int size,
                                 float *coeffs, HeaacIcs *ics, HeaacToolsFrame *tools,
                                 HeaacAacElementInfo *elem, HeaacAacFrameInfo *info)
{
    return heaac_aac_parse_frame_layout_ex(cfg, layout, st, au, size, coeffs, ics, tools, elem, NULL, info);
}

/* What the walk over a layout's access unit leaves for the caller to judge a refusal by (unit_refused above, per
 * element): the window histories in work, the progress of every element that was completed, of the one the refusal
 * stands in, and of the loop around them. */
typedef struct {
    Bits b;
    WinInfo w[HEAAC_MAX_ELEMENTS][2];
    Progress done[HEAAC_MAX_ELEMENTS], at, loop;
    int at_slot, n_cce;
} LayoutWalk;

static int layout_walk(const HeaacAacConfig *cfg, HeaacAacLayout *layout, HeaacAacStream *st, const uint8_t *au, int size,
                       float *coeffs, HeaacIcs *ics, HeaacToolsFrame *tools, HeaacAacElementInfo *elem,
                       const HeaacCceOut *cce, HeaacAacFrameInfo *info, LayoutWalk *lw)
{
    Bits *b = &lw->b;
    bits_init(b, au, size);
    if (peek(b, 12) == 0xfff) {
        HeaacAdtsHeader h;
        const int hs = heaac_adts_parse_header(&h, au, size);
        if (hs < 0) return HEAAC_PARSE_ERR_DATA;
        if (h.num_aac_frames != 1) return HEAAC_PARSE_ERR_UNSUPPORTED;
        if (h.sampling_index != cfg->sampling_index || h.object_type != cfg->object_type) return HEAAC_PARSE_ERR_DATA;
        b->pos = hs * 8;
    }
    const int ne = layout->n_elements;
    WinInfo (*w)[2] = lw->w;
    for (int e = 0; e < ne; e++)
        for (int c = 0; c < 2; c++) {
            w[e][c].window_sequence[0] = st[e].window_sequence[c];
            w[e][c].use_kb_window[0] = st[e].use_kb_window[c];
            w[e][c].window_sequence[1] = w[e][c].use_kb_window[1] = 0;
        }
    memset(elem, 0, (size_t)ne * sizeof(*elem));
    for (int e = 0; e < ne; e++) elem[e].sbr_payload_bit = -1;
    uint8_t seen[4][16];
    memset(seen, 0, sizeof(seen));
    int n_seen = 0, prev_slot = -1, last_cce = -1, prev_type = TYPE_END, type, r;      /* last_cce: the channel element last seen is coupling slot k */
    /* coupling elements: slot k of the layout's list; lists[k] until the output elements are all known */
    WinInfo wc[HEAAC_MAX_CCE];
    HeaacCceFrame cbase[HEAAC_MAX_CCE];
    CceLists lists[HEAAC_MAX_CCE];
    int n_cce = 0;
    memset(cbase, 0, sizeof(cbase));
    if (cce) memset(cce->cce, 0, (size_t)ne * HEAAC_MAX_CCE * sizeof(HeaacCceFrame));
    if (cce && cce->elem) {
        memset(cce->elem, 0, HEAAC_MAX_CCE * sizeof(*cce->elem));
        for (int k = 0; k < HEAAC_MAX_CCE; k++) cce->elem[k].sbr_payload_bit = -1;
    }
    while ((type = (int)bits(b, 3)) != TYPE_END) {
        int tag = (int)bits(b, 4);
        int slot = -1;
        switch (type) {
        case TYPE_SCE:
        case TYPE_CPE:
        case TYPE_LFE: {
            slot = layout_find(layout, seen, type, &tag);
            if (slot < 0) return REF_FAIL(&lw->loop, HEAAC_PARSE_ERR_DATA);    /* "channel element %d.%d is not allocated" */
            if (slot >= ne) return HEAAC_PARSE_ERR_ARG;                /* a layout record not made by the layout functions */
            /* the element decodes as what the bitstream says it is; a pair needs a pair's slot */
            if ((type == TYPE_CPE) != (layout->elem[slot].channels == 2)) return HEAAC_PARSE_ERR_DATA;
            HeaacToolsFrame *t = &tools[slot];
            memset(t, 0, sizeof(*t));
            float *co = coeffs + (size_t)slot * 2048;
            elem[slot].type = (uint8_t)type;
            elem[slot].tag = (uint8_t)tag;
            elem[slot].seq = (uint8_t)n_seen;
            lw->at_slot = slot;                        /* the element a refusal from here on stands in */
            memset(&lw->at, 0, sizeof(lw->at));
            if (type == TYPE_CPE) r = read_cpe(cfg, b, t, w[slot], co, &lw->at);
            else r = read_ics(cfg, b, 0, &t->ch[0], &w[slot][0], co, &lw->at, 0);
            if (r < 0) return r;
            lw->done[slot] = lw->at;
            lw->at_slot = -1;
            elem[slot].present = 1;
            n_seen++;
            break;
        }
        case TYPE_CCE: {
            /* get_che: a tag met twice moves up; only what a program config element named is allocated */
            while (tag < 16 && seen[TYPE_CCE][tag]) tag++;
            if (tag == 16) return HEAAC_PARSE_ERR_DATA;
            seen[TYPE_CCE][tag] = 1;
            const int k = layout->tag_map[TYPE_CCE][tag] - 1;
            if (k < 0) return REF_FAIL(&lw->loop, HEAAC_PARSE_ERR_DATA);       /* "channel element 2.%d is not allocated" */
            if (!cce || k >= HEAAC_MAX_CCE) return HEAAC_PARSE_ERR_UNSUPPORTED;
            wc[k].window_sequence[0] = st[0].cce_window_sequence[tag];
            wc[k].use_kb_window[0] = st[0].cce_use_kb_window[tag];
            wc[k].window_sequence[1] = wc[k].use_kb_window[1] = 0;
            memset(&cce->tools[k], 0, sizeof(HeaacToolsFrame));
            r = read_cce(cfg, b, tag, &cbase[k], &lists[k], &cce->tools[k].ch[0], &wc[k], cce->coeffs + k * 1024);
            if (r < 0) return r;
            cbase[k].outputs_before = (uint8_t)n_seen;
            cbase[k].seq = (uint8_t)n_cce++;
            lw->n_cce = n_cce;
            last_cce = k;
            if (cce->elem) {
                cce->elem[k].present = 1;
                cce->elem[k].type = TYPE_CCE;
                cce->elem[k].tag = (uint8_t)tag;
                cce->elem[k].seq = cbase[k].seq;
            }
            break;
        }
        case TYPE_DSE:
            if ((r = skip_dse(b)) < 0) return r;
            break;
        case TYPE_PCE:
            if ((r = skip_pce(b)) < 0) return r;
            break;
        case TYPE_FIL: {
            int cnt = tag;
            if (cnt == 15) cnt += (int)bits(b, 8) - 1;
            if (bits_left(b) < 8 * cnt) return REF_OVERREAD(&lw->loop);              /* :2053-2056 */
            int at, bytes = 0, crc = 0;
            read_fil(b, cnt, &at, &bytes, &crc);
            if (at >= 0) {
                /* decode_extension_payload (:1650-1690) hands the payload to the channel element last seen, and to its
                 * SBR reader the type of the element directly in front (:2059): anything but that element itself and
                 * the reader switches the element's SBR off (aacsbr.c:996-1000) -- so it does for an LFE */
                if (prev_slot < 0 && last_cce < 0) return HEAAC_PARSE_ERR_DATA;   /* "SBR was found before the first channel element" */
                HeaacAacElementInfo *to;
                if (last_cce >= 0) {
                    /* the coupling element's own SBR (it goes through ff_sbr_apply when it couples AFTER_IMDCT, :1924) */
                    if (!cce->elem) return HEAAC_PARSE_ERR_UNSUPPORTED;
                    to = &cce->elem[last_cce];
                } else {
                    to = &elem[prev_slot];
                }
                if (to->sbr_payload_bit >= 0) return HEAAC_PARSE_ERR_UNSUPPORTED;  /* a second payload for the element */
                to->sbr_payload_bit = at;
                to->sbr_payload_bytes = bytes;
                to->sbr_crc = (uint8_t)crc;
                to->sbr_misplaced = (uint8_t)(prev_type != TYPE_SCE && prev_type != TYPE_CPE && prev_type != TYPE_CCE);
            }
            break;
        }
        default:
            return HEAAC_PARSE_ERR_UNSUPPORTED;
        }
        if (slot >= 0) prev_slot = slot;
        if (type < TYPE_DSE && type != TYPE_CCE) last_cce = -1;
        prev_type = type;
        if (b->over) return HEAAC_PARSE_ERR_OVERREAD;
        if (bits_left(b) < 3) return REF_OVERREAD(&lw->loop);                         /* :2072-2075 */
    }
    if (!n_seen) return HEAAC_PARSE_ERR_DATA;
    /* every coupling element against every output element: apply_channel_coupling compares the target list with the
     * element's place in ac->che[type][] (:1903-1933 hands it `i`), which a program config element makes its tag */
    for (int k = 0; k < HEAAC_MAX_CCE; k++) {
        if (!cbase[k].present) continue;
        for (int e = 0; e < ne; e++) {
            HeaacCceFrame *o = &cce->cce[e * HEAAC_MAX_CCE + k];
            *o = cbase[k];
            o->behind_target = (uint8_t)(elem[e].present && cbase[k].outputs_before > elem[e].seq);
            if ((r = cce_resolve(&lists[k], layout->elem[e].type, layout->elem[e].id, o)) < 0) return r;
        }
    }
    for (int k = 0; k < HEAAC_MAX_CCE; k++) {
        if (!cbase[k].present) continue;
        cce->ics[k].window_sequence[0] = wc[k].window_sequence[0];
        cce->ics[k].window_sequence[1] = wc[k].window_sequence[1];
        cce->ics[k].use_kb_window[0] = wc[k].use_kb_window[0];
        cce->ics[k].use_kb_window[1] = wc[k].use_kb_window[1];
        st[0].cce_window_sequence[cbase[k].elem_id] = wc[k].window_sequence[0];
        st[0].cce_use_kb_window[cbase[k].elem_id] = wc[k].use_kb_window[0];
    }
    for (int e = 0; e < ne; e++) {
        if (!elem[e].present) continue;
        for (int c = 0; c < layout->elem[e].channels; c++) {
            HeaacIcs *o = &ics[e * 2 + c];
            o->window_sequence[0] = w[e][c].window_sequence[0];
            o->window_sequence[1] = w[e][c].window_sequence[1];
            o->use_kb_window[0] = w[e][c].use_kb_window[0];
            o->use_kb_window[1] = w[e][c].use_kb_window[1];
            st[e].window_sequence[c] = w[e][c].window_sequence[0];
            st[e].use_kb_window[c] = w[e][c].use_kb_window[0];
        }
    }
    if (info) {
        memset(info, 0, sizeof(*info));
        info->channels = layout->channels;
        info->bits_consumed = b->pos;
        info->sbr_payload_bit = -1;
        info->n_cce = n_cce;
    }
    return HEAAC_PARSE_OK;
}



int heaac_aac_parse_frame_layout_ex(const HeaacAacConfig *cfg, HeaacAacLayout *layout, HeaacAacStream *st,
                                    const uint8_t *au, int size,
                                    float *coeffs, HeaacIcs *ics, HeaacToolsFrame *tools,
                                    HeaacAacElementInfo *elem, const HeaacCceOut *cce, HeaacAacFrameInfo *info)
{
    if (!cfg || !layout || !st || !au || size <= 0 || !coeffs || !ics || !tools || !elem ||
        cfg->sampling_index < 0 || cfg->sampling_index > 12 ||
        layout->n_elements < 1 || layout->n_elements > HEAAC_MAX_ELEMENTS ||
        (cce && (!cce->cce || !cce->coeffs || !cce->ics || !cce->tools)))
        return HEAAC_PARSE_ERR_ARG;
    pthread_once(&g_once, tables_init);
    if (g_tables_bad) return HEAAC_PARSE_ERR_ARG;
    LayoutWalk lw;
    memset(&lw, 0, sizeof(lw));
    lw.at_slot = -1;
    const int r = layout_walk(cfg, layout, st, au, size, coeffs, ics, tools, elem, cce, info, &lw);
    if (r >= 0) return r;
    /* Refused.  As for a one-element stream (unit_refused): where the refusal is the reference's own and no coupling
     * element has been read, the elements completed before it keep what their decoders did -- window history moved,
     * noise drawn, predictors stepped: their records stand, `present` and `seq` say which and in which order -- and
     * the element the refusal stands in is rewritten into records that do as much as its decoder had done. */
    HeaacAacFrameInfo fi = { 0, 0, -1, 0, 0, 0, 0, 0, 0 };
    const Progress *why = lw.at_slot >= 0 ? &lw.at : &lw.loop;
    if (why->as_reference && (!lw.b.over || why->ref_overread) && !lw.n_cce) {
        fi.refused = HEAAC_REFUSED_AS_REFERENCE;
        int work = 0;
        const int main_profile = cfg->object_type == HEAAC_AOT_AAC_MAIN;
        for (int e = 0; e < layout->n_elements; e++) {
            if (!elem[e].present) continue;
            for (int c = 0; c < layout->elem[e].channels; c++) {
                st[e].window_sequence[c] = lw.w[e][c].window_sequence[0];
                st[e].use_kb_window[c] = lw.w[e][c].use_kb_window[0];
                work |= main_profile || noise_draws(&tools[e].ch[c], 128);
            }
        }
        if (lw.at_slot >= 0) {
            const int e = lw.at_slot;
            HeaacAacFrameInfo part;
            unit_refused(cfg, &st[e], &lw.b, lw.w[e], &lw.at, 0, 2, coeffs + (size_t)e * 2048, &tools[e], &part);
            if (part.refused & HEAAC_REFUSED_RUN_TOOLS) {
                elem[e].present = 1;                   /* (type, tag and seq were set when the element began) */
                work = 1;
            }
        }
        if (work) fi.refused |= HEAAC_REFUSED_RUN_TOOLS;
    }
    if (info) *info = fi;
    return r;
}

/* ------------------------------------------------------------------------------------------ */
/* batch over streams                                                                            */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    const HeaacAacConfig *cfg; HeaacAacStream *st; const uint8_t *const *au; const int *size;
    float *coeffs; HeaacIcs *ics; HeaacToolsFrame *tools; HeaacAacFrameInfo *info; int *status;
    size_t lo, hi; int failed;
} Job;

static void *job_run(void *p)
{
    Job *j = (Job *)p;
    for (size_t f = j->lo; f < j->hi; f++) {
        const int r = heaac_aac_parse_frame(j->cfg, &j->st[f], j->au[f], j->size[f], j->coeffs + f * 2048,
                                            j->ics + f * 2, j->tools + f, j->info ? j->info + f : NULL);
        if (j->status) j->status[f] = r;
        j->failed += r != HEAAC_PARSE_OK;
    }
    return NULL;
}

int heaac_aac_parse_batch(const HeaacAacConfig *cfg, HeaacAacStream *st,
                          const uint8_t *const *au, const int *size, size_t n,
                          float *coeffs, HeaacIcs *ics, HeaacToolsFrame *tools,
                          HeaacAacFrameInfo *info, int *status, int threads)
{
    if (!cfg || !st || !au || !size || !coeffs || !ics || !tools) return HEAAC_PARSE_ERR_ARG;
    if (!n) return 0;
    pthread_once(&g_once, tables_init);
    if (g_tables_bad) return HEAAC_PARSE_ERR_ARG;
    if (threads <= 0) threads = (int)sysconf(_SC_NPROCESSORS_ONLN);
    if (threads < 1) threads = 1;
    if ((size_t)threads > n) threads = (int)n;
    if (threads > 256) threads = 256;
    Job jobs[256];
    pthread_t tid[256];
    for (int t = 0; t < threads; t++) {
        Job j = { cfg, st, au, size, coeffs, ics, tools, info, status, n * t / threads, n * (t + 1) / threads, 0 };
        jobs[t] = j;
    }
    int started = 0;
    for (int t = 1; t < threads; t++) {
        if (pthread_create(&tid[t], NULL, job_run, &jobs[t]) != 0) break;
        started = t;
    }
    job_run(&jobs[0]);
    for (int t = started + 1; t < threads; t++) job_run(&jobs[t]);     /* threads that could not start: inline */
    int failed = jobs[0].failed;
    for (int t = 1; t < threads; t++) {
        if (t <= started) pthread_join(tid[t], NULL);
        failed += jobs[t].failed;
    }
    return failed;
}
