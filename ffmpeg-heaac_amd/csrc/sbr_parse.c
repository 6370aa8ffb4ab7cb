/* sbr_parse.c -- host-side parser of the SBR extension payload and its Parametric Stereo data
 * (include/heaac_parse.h, second slice).
 *
 * Own structure: the stream state is a plain record (HeaacSbrStream) the caller owns, one per stream;
 * derived band tables are never kept per stream -- a header that changes the tables is looked up in (or
 * added to) a table shared by the whole batch and the frame records carry its index, which is what
 * heaac_he_decode_batch takes.  Huffman codes are walked through binary trees built once from the ISO
 * (code, length) pairs of sbr_iso_tables.h.
 *
 * The bit order and every value follow the reference: ff_decode_sbr_extension (aacsbr.c:1044-1090),
 * read_sbr_header (:207-262), read_sbr_grid (:609-745), copy_sbr_grid (:747-766), read_sbr_dtdf /
 * read_sbr_invf / read_sbr_envelope / read_sbr_noise (:768-898), read_sbr_extension (:900-926),
 * read_sbr_single_channel_element / read_sbr_channel_pair_element / read_sbr_data (:928-1020),
 * ff_ps_read_data with its parameter readers (aacps.c:84-279).
 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "heaac_parse.h"
#include "sbr_iso_tables.h"
#include "parse_bits.h"
#include "validate.h"

/* table order of sbr_iso_tables.h = the reference's enums (aacsbr.c:45-56, aacps.c:50-61) */
enum { T_ENV_15, F_ENV_15, T_ENV_BAL_15, F_ENV_BAL_15, T_ENV_30, F_ENV_30, T_ENV_BAL_30, F_ENV_BAL_30,
       T_NOISE_30, T_NOISE_BAL_30 };
enum { PS_IID_DF1, PS_IID_DT1, PS_IID_DF0, PS_IID_DT0, PS_ICC_DF, PS_ICC_DT, PS_IPD_DF, PS_IPD_DT,
       PS_OPD_DF, PS_OPD_DT };
enum { FIXFIX, FIXVAR, VARFIX, VARVAR };
enum { EXTENSION_ID_PS = 2 };

static Tree g_sbr_tree[10], g_ps_tree[10];
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static void tables_init(void)
{
    for (int t = 0; t < 10; t++) {
        tree_build(&g_sbr_tree[t], sbr_huff_code + sbr_huff_first[t], NULL, sbr_huff_bits + sbr_huff_first[t],
                   sbr_huff_first[t + 1] - sbr_huff_first[t]);
        tree_build(&g_ps_tree[t], ps_huff_code + ps_huff_first[t], NULL, ps_huff_bits + ps_huff_first[t],
                   ps_huff_first[t + 1] - ps_huff_first[t]);
    }
}

uint64_t heaac_sbr_tables_fingerprint(void)
{
    uint64_t h = 1469598103934665603ull;
#define MIX(arr) do { const uint8_t *p_ = (const uint8_t *)(arr); for (size_t i_ = 0; i_ < sizeof(arr); i_++) { h ^= p_[i_]; h *= 1099511628211ull; } } while (0)
    MIX(sbr_huff_first); MIX(sbr_huff_code); MIX(sbr_huff_bits); MIX(sbr_huff_lav);
    MIX(ps_huff_first); MIX(ps_huff_code); MIX(ps_huff_bits); MIX(ps_huff_offset);
#undef MIX
    return h;
}

/* ------------------------------------------------------------------------------------------ */
/* header table                                                                                  */
/* ------------------------------------------------------------------------------------------ */
typedef struct { int32_t v[13]; } HdrKey;            /* SBR rate + the 12 header fields */

struct HeaacSbrHeaderTable {
    HeaacSbrHeader *h;
    HdrKey *key;
    size_t cap;
    volatile size_t n;
    pthread_mutex_t lock;
};

HeaacSbrHeaderTable *heaac_sbr_table_create(size_t capacity)
{
    if (capacity < 1 || capacity > 65535) return NULL;
    HeaacSbrHeaderTable *t = (HeaacSbrHeaderTable *)calloc(1, sizeof(*t));
    if (!t) return NULL;
    t->h = (HeaacSbrHeader *)calloc(capacity, sizeof(HeaacSbrHeader));
    t->key = (HdrKey *)calloc(capacity, sizeof(HdrKey));
    if (!t->h || !t->key) { free(t->h); free(t->key); free(t); return NULL; }
    t->cap = capacity;
    pthread_mutex_init(&t->lock, NULL);
    /* entry 0: no header yet */
    t->h[0].kx = 32;
    memset(t->h[0].map_hi, 0xff, 6 * 64);
    for (int i = 0; i < 13; i++) t->key[0].v[i] = -1;
    t->n = 1;
    return t;
}

void heaac_sbr_table_destroy(HeaacSbrHeaderTable *t)
{
    if (!t) return;
    pthread_mutex_destroy(&t->lock);
    free(t->h); free(t->key); free(t);
}

size_t heaac_sbr_table_count(const HeaacSbrHeaderTable *t) { return t ? t->n : 0; }
const HeaacSbrHeader *heaac_sbr_table_data(const HeaacSbrHeaderTable *t) { return t ? t->h : NULL; }

/* index of the header built from `k`, -1: the tables cannot be built (sbr_reset fails), -2: table full */
static int table_find_or_add(HeaacSbrHeaderTable *t, const HdrKey *k)
{
    int idx = -1;
    pthread_mutex_lock(&t->lock);
    for (size_t i = 1; i < t->n; i++)
        if (!memcmp(&t->key[i], k, sizeof(*k))) { idx = (int)i; break; }
    if (idx < 0) {
        if (t->n >= t->cap) {
            idx = -2;
        } else {
            HeaacSbrHeader h;
            const int32_t *v = k->v;
            /* a header the reference builds but whose tables the decode kernels cannot take (no limiter band
             * left over a dropped patch: the reference then reads gains of earlier frames) counts as not built */
            if (heaac_sbr_make_header(&h, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9], v[10], v[11]) == 0 &&
                heaac_check_sbr_header(&h) == HEAAC_BAD_NONE) {
                t->h[t->n] = h;
                t->key[t->n] = *k;
                __sync_synchronize();                 /* the entry is complete before it becomes visible */
                idx = (int)t->n;
                t->n = t->n + 1;
            }
        }
    }
    pthread_mutex_unlock(&t->lock);
    return idx;
}

static void make_key(HdrKey *k, const HeaacSbrStream *s, int sbr_rate)
{
    k->v[0] = sbr_rate;
    k->v[1] = s->bs_start_freq;  k->v[2] = s->bs_stop_freq;   k->v[3] = s->bs_xover_band;
    k->v[4] = s->bs_freq_scale;  k->v[5] = s->bs_alter_scale; k->v[6] = s->bs_noise_bands;
    k->v[7] = s->bs_limiter_bands; k->v[8] = s->bs_limiter_gains;
    k->v[9] = s->bs_interpol_freq; k->v[10] = s->bs_smoothing_mode; k->v[11] = s->bs_amp_res_header;
    k->v[12] = 0;
}

/* ------------------------------------------------------------------------------------------ */
/* stream state                                                                                  */
/* ------------------------------------------------------------------------------------------ */
void heaac_sbr_stream_init(HeaacSbrStream *st, size_t n)
{
    memset(st, 0, n * sizeof(*st));
    for (size_t i = 0; i < n; i++) {
        st[i].kx[0] = st[i].kx[1] = 32;               /* ff_aac_sbr_ctx_init, aacsbr.c:130-131 */
        st[i].data[0].e_a[1] = st[i].data[1].e_a[1] = -1;
    }
}

size_t heaac_sbr_stream_bytes(void) { return sizeof(HeaacSbrStream); }

/* ------------------------------------------------------------------------------------------ */
/* SBR header (read_sbr_header, aacsbr.c:207-262).  Returns 1 if the band tables must be rebuilt. */
/* ------------------------------------------------------------------------------------------ */
static int read_header(HeaacSbrStream *s, Bits *b, int *tables_touched)
{
    const uint8_t old[6] = { s->bs_start_freq, s->bs_stop_freq, s->bs_xover_band,
                             s->bs_freq_scale, s->bs_alter_scale, s->bs_noise_bands };
    const uint8_t old_rest[5] = { s->bs_limiter_bands, s->bs_limiter_gains, s->bs_interpol_freq,
                                  s->bs_smoothing_mode, s->bs_amp_res_header };
    s->start = 1;
    s->bs_amp_res_header = (uint8_t)bit1(b);
    s->bs_start_freq     = (uint8_t)bits(b, 4);
    s->bs_stop_freq      = (uint8_t)bits(b, 4);
    s->bs_xover_band     = (uint8_t)bits(b, 3);
    bits(b, 2);                                        /* bs_reserved */
    const unsigned extra_1 = bit1(b), extra_2 = bit1(b);
    if (extra_1) {
        s->bs_freq_scale  = (uint8_t)bits(b, 2);
        s->bs_alter_scale = (uint8_t)bit1(b);
        s->bs_noise_bands = (uint8_t)bits(b, 2);
    } else {
        s->bs_freq_scale = 2; s->bs_alter_scale = 1; s->bs_noise_bands = 2;
    }
    const uint8_t now[6] = { s->bs_start_freq, s->bs_stop_freq, s->bs_xover_band,
                             s->bs_freq_scale, s->bs_alter_scale, s->bs_noise_bands };
    const int reset = !s->have_spectrum || memcmp(old, now, 6) != 0;
    if (extra_2) {
        s->bs_limiter_bands  = (uint8_t)bits(b, 2);
        s->bs_limiter_gains  = (uint8_t)bits(b, 2);
        s->bs_interpol_freq  = (uint8_t)bit1(b);
        s->bs_smoothing_mode = (uint8_t)bit1(b);
    } else {
        s->bs_limiter_bands = 2; s->bs_limiter_gains = 2; s->bs_interpol_freq = 1; s->bs_smoothing_mode = 1;
    }
    const uint8_t rest[5] = { s->bs_limiter_bands, s->bs_limiter_gains, s->bs_interpol_freq,
                              s->bs_smoothing_mode, s->bs_amp_res_header };
    /* the limiter table (:258-259) and the four scalars live in the derived record: any change of them
     * selects another record without a reset */
    *tables_touched = memcmp(old_rest, rest, 5) != 0;
    return reset;
}

/* ------------------------------------------------------------------------------------------ */
/* time / frequency grid (read_sbr_grid, aacsbr.c:609-745)                                       */
/* ------------------------------------------------------------------------------------------ */
static const int8_t k_ceil_log2[6] = { 0, 1, 2, 2, 3, 3 };

static int read_grid(const HeaacSbrStream *s, Bits *b, HeaacSbrChanState *c)
{
    unsigned pointer = 0;
    int abs_bord_trail = 16, num_rel_lead, num_rel_trail;
    const unsigned num_env_old = c->bs_num_env;

    c->bs_freq_res[0] = c->bs_freq_res[c->bs_num_env];
    c->bs_amp_res = s->bs_amp_res_header;
    c->t_env_num_env_old = c->t_env[num_env_old];

    switch (c->bs_frame_class = (uint8_t)bits(b, 2)) {
    case FIXFIX: {
        const int L = 1 << bits(b, 2);
        if (L == 1) c->bs_amp_res = 0;
        if (L > 4) return HEAAC_PARSE_ERR_DATA;
        c->bs_num_env = (uint8_t)L;
        c->t_env[0] = 0;
        c->t_env[L] = 16;
        const int step = (16 + (L >> 1)) / L;
        for (int i = 0; i < L - 1; i++) c->t_env[i + 1] = (uint8_t)(c->t_env[i] + step);
        c->bs_freq_res[1] = (uint8_t)bit1(b);
        for (int i = 1; i < L; i++) c->bs_freq_res[i + 1] = c->bs_freq_res[1];
        break;
    }
    case FIXVAR: {
        abs_bord_trail += (int)bits(b, 2);
        num_rel_trail = (int)bits(b, 2);
        const int L = num_rel_trail + 1;
        c->bs_num_env = (uint8_t)L;
        c->t_env[0] = 0;
        c->t_env[L] = (uint8_t)abs_bord_trail;
        for (int i = 0; i < num_rel_trail; i++)
            c->t_env[L - 1 - i] = (uint8_t)(c->t_env[L - i] - 2 * bits(b, 2) - 2);
        pointer = bits(b, k_ceil_log2[L]);
        for (int i = 0; i < L; i++) c->bs_freq_res[L - i] = (uint8_t)bit1(b);
        break;
    }
    case VARFIX: {
        c->t_env[0] = (uint8_t)bits(b, 2);
        num_rel_lead = (int)bits(b, 2);
        const int L = num_rel_lead + 1;
        c->bs_num_env = (uint8_t)L;
        c->t_env[L] = 16;
        for (int i = 0; i < num_rel_lead; i++)
            c->t_env[i + 1] = (uint8_t)(c->t_env[i] + 2 * bits(b, 2) + 2);
        pointer = bits(b, k_ceil_log2[L]);
        for (int i = 0; i < L; i++) c->bs_freq_res[1 + i] = (uint8_t)bit1(b);
        break;
    }
    default: { /* VARVAR */
        c->t_env[0] = (uint8_t)bits(b, 2);
        abs_bord_trail += (int)bits(b, 2);
        num_rel_lead  = (int)bits(b, 2);
        num_rel_trail = (int)bits(b, 2);
        const int L = num_rel_lead + num_rel_trail + 1;
        if (L > 5) return HEAAC_PARSE_ERR_DATA;
        c->bs_num_env = (uint8_t)L;
        c->t_env[L] = (uint8_t)abs_bord_trail;
        for (int i = 0; i < num_rel_lead; i++)
            c->t_env[i + 1] = (uint8_t)(c->t_env[i] + 2 * bits(b, 2) + 2);
        for (int i = 0; i < num_rel_trail; i++)
            c->t_env[L - 1 - i] = (uint8_t)(c->t_env[L - i] - 2 * bits(b, 2) - 2);
        pointer = bits(b, k_ceil_log2[L]);
        for (int i = 0; i < L; i++) c->bs_freq_res[1 + i] = (uint8_t)bit1(b);
        break;
    }
    }
    const int L = c->bs_num_env;
    if (pointer > (unsigned)L + 1) return HEAAC_PARSE_ERR_DATA;
    /* the reference rejects t_env[i-1] > t_env[i] (:715-720); equality is rejected here too (see the
     * header).  uint8 wrap of a trailing border walked below zero shows up as a non-monotone table. */
    for (int i = 1; i <= L; i++)
        if (c->t_env[i - 1] >= c->t_env[i]) return HEAAC_PARSE_ERR_DATA;

    c->bs_num_noise = (uint8_t)((L > 1) + 1);
    c->t_q[0] = c->t_env[0];
    c->t_q[c->bs_num_noise] = c->t_env[L];
    if (c->bs_num_noise > 1) {
        unsigned idx;
        if (c->bs_frame_class == FIXFIX) {
            idx = (unsigned)L >> 1;
        } else if (c->bs_frame_class & 1) {            /* FIXVAR, VARVAR */
            const int p1 = (int)pointer - 1;
            idx = (unsigned)(L - (p1 > 1 ? p1 : 1));
        } else {                                       /* VARFIX */
            if (!pointer)          idx = 1;
            else if (pointer == 1) idx = (unsigned)L - 1;
            else                   idx = pointer - 1;
        }
        c->t_q[1] = c->t_env[idx];
    }
    c->e_a[0] = (int8_t)-(c->e_a[1] != (int)num_env_old);
    c->e_a[1] = -1;
    if ((c->bs_frame_class & 1) && pointer)
        c->e_a[1] = (int8_t)(L + 1 - (int)pointer);
    else if (c->bs_frame_class == VARFIX && pointer > 1)
        c->e_a[1] = (int8_t)(pointer - 1);
    return HEAAC_PARSE_OK;
}

/* copy_sbr_grid (:747-766) */
static void copy_grid(HeaacSbrChanState *dst, const HeaacSbrChanState *src)
{
    dst->bs_freq_res[0]    = dst->bs_freq_res[dst->bs_num_env];
    dst->t_env_num_env_old = dst->t_env[dst->bs_num_env];
    dst->e_a[0]            = (int8_t)-(dst->e_a[1] != dst->bs_num_env);
    memcpy(dst->bs_freq_res + 1, src->bs_freq_res + 1, 7);
    memcpy(dst->t_env, src->t_env, sizeof(dst->t_env));
    memcpy(dst->t_q, src->t_q, sizeof(dst->t_q));
    dst->bs_num_env     = src->bs_num_env;
    dst->bs_amp_res     = src->bs_amp_res;
    dst->bs_num_noise   = src->bs_num_noise;
    dst->bs_frame_class = src->bs_frame_class;
    dst->e_a[1]         = src->e_a[1];
}

static void read_dtdf(Bits *b, HeaacSbrChanState *c)
{
    for (int i = 0; i < c->bs_num_env; i++)   c->bs_df_env[i]   = (uint8_t)bit1(b);
    for (int i = 0; i < c->bs_num_noise; i++) c->bs_df_noise[i] = (uint8_t)bit1(b);
}

static void read_invf(const HeaacSbrHeader *h, Bits *b, HeaacSbrChanState *c)
{
    memcpy(c->bs_invf_mode[1], c->bs_invf_mode[0], 5);
    for (int i = 0; i < h->n_q; i++) c->bs_invf_mode[0][i] = (uint8_t)bits(b, 2);
}

static inline int huff(Bits *b, int table, int *bad)
{
    const int s = tree_read(&g_sbr_tree[table], b);
    if (s < 0) { *bad = 1; return 0; }                /* complete codes: reachable only past the end */
    return s - sbr_huff_lav[table];
}

/* read_sbr_envelope (:783-858) */
static void read_envelope(const HeaacSbrStream *s, const HeaacSbrHeader *h, Bits *b, HeaacSbrChanState *c, int ch, int *bad)
{
    int start_bits, t_huff, f_huff;
    const int delta = (ch == 1 && s->bs_coupling == 1) + 1;
    const int odd = h->n[1] & 1;

    if (s->bs_coupling && ch) {
        if (c->bs_amp_res) { start_bits = 5; t_huff = T_ENV_BAL_30; f_huff = F_ENV_BAL_30; }
        else               { start_bits = 6; t_huff = T_ENV_BAL_15; f_huff = F_ENV_BAL_15; }
    } else {
        if (c->bs_amp_res) { start_bits = 6; t_huff = T_ENV_30; f_huff = F_ENV_30; }
        else               { start_bits = 7; t_huff = T_ENV_15; f_huff = F_ENV_15; }
    }
    for (int i = 0; i < c->bs_num_env; i++) {
        const int res = c->bs_freq_res[i + 1], nb = h->n[res];
        int32_t *cur = c->env_facs[i + 1];
        const int32_t *prev = c->env_facs[i];
        if (c->bs_df_env[i]) {
            if (res == c->bs_freq_res[i]) {
                for (int j = 0; j < nb; j++) cur[j] = prev[j] + delta * huff(b, t_huff, bad);
            } else if (res) {
                for (int j = 0; j < nb; j++) cur[j] = prev[(j + odd) >> 1] + delta * huff(b, t_huff, bad);
            } else {
                for (int j = 0; j < nb; j++) cur[j] = prev[j ? 2 * j - odd : 0] + delta * huff(b, t_huff, bad);
            }
        } else {
            cur[0] = delta * (int)bits(b, start_bits);
            for (int j = 1; j < nb; j++) cur[j] = cur[j - 1] + delta * huff(b, f_huff, bad);
        }
    }
    memcpy(c->env_facs[0], c->env_facs[c->bs_num_env], sizeof(c->env_facs[0]));
}

/* read_sbr_noise (:860-898) */
static void read_noise(const HeaacSbrStream *s, const HeaacSbrHeader *h, Bits *b, HeaacSbrChanState *c, int ch, int *bad)
{
    const int delta = (ch == 1 && s->bs_coupling == 1) + 1;
    const int t_huff = (s->bs_coupling && ch) ? T_NOISE_BAL_30 : T_NOISE_30;
    const int f_huff = (s->bs_coupling && ch) ? F_ENV_BAL_30 : F_ENV_30;
    for (int i = 0; i < c->bs_num_noise; i++) {
        int32_t *cur = c->noise_facs[i + 1];
        const int32_t *prev = c->noise_facs[i];
        if (c->bs_df_noise[i]) {
            for (int j = 0; j < h->n_q; j++) cur[j] = prev[j] + delta * huff(b, t_huff, bad);
        } else {
            cur[0] = delta * (int)bits(b, 5);
            for (int j = 1; j < h->n_q; j++) cur[j] = cur[j - 1] + delta * huff(b, f_huff, bad);
        }
    }
    memcpy(c->noise_facs[0], c->noise_facs[c->bs_num_noise], sizeof(c->noise_facs[0]));
}

static void read_harmonics(const HeaacSbrHeader *h, Bits *b, HeaacSbrChanState *c)
{
    if ((c->bs_add_harmonic_flag = (uint8_t)bit1(b)))
        for (int i = 0; i < h->n[1]; i++) c->bs_add_harmonic[i] = (uint8_t)bit1(b);
}

/* the uint8 range of the frame record */
static int facs_in_range(const HeaacSbrHeader *h, const HeaacSbrChanState *c)
{
    for (int e = 1; e <= c->bs_num_env; e++)
        for (int k = 0; k < h->n[c->bs_freq_res[e]]; k++)
            if (c->env_facs[e][k] < 0 || c->env_facs[e][k] > 255) return 0;
    for (int e = 1; e <= c->bs_num_noise; e++)
        for (int k = 0; k < h->n_q; k++)
            if (c->noise_facs[e][k] < 0 || c->noise_facs[e][k] > 255) return 0;
    return 1;
}

static int read_sce(HeaacSbrStream *s, const HeaacSbrHeader *h, Bits *b)
{
    int bad = 0;
    HeaacSbrChanState *c = &s->data[0];
    if (bit1(b)) bits(b, 4);                           /* bs_data_extra, bs_reserved */
    if (read_grid(s, b, c)) return HEAAC_PARSE_ERR_DATA;
    read_dtdf(b, c);
    read_invf(h, b, c);
    read_envelope(s, h, b, c, 0, &bad);
    read_noise(s, h, b, c, 0, &bad);
    read_harmonics(h, b, c);
    return (bad || !facs_in_range(h, c)) ? HEAAC_PARSE_ERR_DATA : HEAAC_PARSE_OK;
}

static int read_cpe(HeaacSbrStream *s, const HeaacSbrHeader *h, Bits *b)
{
    int bad = 0;
    HeaacSbrChanState *c0 = &s->data[0], *c1 = &s->data[1];
    if (bit1(b)) bits(b, 8);
    if ((s->bs_coupling = (uint8_t)bit1(b))) {
        if (read_grid(s, b, c0)) return HEAAC_PARSE_ERR_DATA;
        copy_grid(c1, c0);
        read_dtdf(b, c0);
        read_dtdf(b, c1);
        read_invf(h, b, c0);
        memcpy(c1->bs_invf_mode[1], c1->bs_invf_mode[0], 5);
        memcpy(c1->bs_invf_mode[0], c0->bs_invf_mode[0], 5);
        read_envelope(s, h, b, c0, 0, &bad);
        read_noise(s, h, b, c0, 0, &bad);
        read_envelope(s, h, b, c1, 1, &bad);
        read_noise(s, h, b, c1, 1, &bad);
    } else {
        if (read_grid(s, b, c0) || read_grid(s, b, c1)) return HEAAC_PARSE_ERR_DATA;
        read_dtdf(b, c0);
        read_dtdf(b, c1);
        read_invf(h, b, c0);
        read_invf(h, b, c1);
        read_envelope(s, h, b, c0, 0, &bad);
        read_envelope(s, h, b, c1, 1, &bad);
        read_noise(s, h, b, c0, 0, &bad);
        read_noise(s, h, b, c1, 1, &bad);
    }
    read_harmonics(h, b, c0);
    read_harmonics(h, b, c1);
    return (bad || !facs_in_range(h, c0) || !facs_in_range(h, c1)) ? HEAAC_PARSE_ERR_DATA : HEAAC_PARSE_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* Parametric Stereo (ff_ps_read_data, aacps.c:150-279)                                          */
/* ------------------------------------------------------------------------------------------ */
static const int8_t k_num_env_tab[2][4] = { { 0, 1, 2, 4 }, { 1, 2, 3, 4 } };
static const int8_t k_nr_iidicc_par[6] = { 10, 20, 34, 10, 20, 34 };
static const int8_t k_nr_ipdopd_par[6] = { 5, 11, 17, 5, 11, 17 };
static const int8_t k_huff_iid[4] = { PS_IID_DF0, PS_IID_DF1, PS_IID_DT0, PS_IID_DT1 };

enum { PAR_IID, PAR_ICC, PAR_IPDOPD };

/* READ_PAR_DATA (aacps.c:84-119): kind selects offset, mask and the error condition */
static int read_par(Bits *b, HeaacPsState *ps, int8_t (*par)[34], int kind, int num, int table, int e, int dt)
{
    const int offset = kind == PAR_IPDOPD ? 0 : ps_huff_offset[table];
    const int8_t *prev = NULL;
    if (dt) {
        int e_prev = e ? e - 1 : ps->num_env_old - 1;
        if (e_prev < 0) e_prev = 0;
        prev = par[e_prev];
    }
    int val = 0;
    for (int k = 0; k < num; k++) {
        const int sym = tree_read(&g_ps_tree[table], b);
        if (sym < 0) return HEAAC_PARSE_ERR_DATA;
        if (dt) val = prev[k] + sym - offset;
        else    val += sym - offset;
        if (kind == PAR_IPDOPD) val &= 7;
        par[e][k] = (int8_t)val;
        if (kind == PAR_IID && abs(par[e][k]) > 7 + 8 * ps->iid_quant) return HEAAC_PARSE_ERR_DATA;
        if (kind == PAR_ICC && (par[e][k] < 0 || par[e][k] > 7)) return HEAAC_PARSE_ERR_DATA;
    }
    return HEAAC_PARSE_OK;
}

/* ps_read_extension_data (:121-139): bits consumed */
static int read_ps_extension(Bits *b, HeaacPsState *ps, int id)
{
    const int at = b->pos;
    if (id) return 0;
    ps->enable_ipdopd = (uint8_t)bit1(b);
    if (ps->enable_ipdopd) {
        for (int e = 0; e < ps->num_env; e++) {
            int dt = (int)bit1(b);
            read_par(b, ps, ps->ipd_par, PAR_IPDOPD, ps->nr_ipdopd_par, dt ? PS_IPD_DT : PS_IPD_DF, e, dt);
            dt = (int)bit1(b);
            read_par(b, ps, ps->opd_par, PAR_IPDOPD, ps->nr_ipdopd_par, dt ? PS_OPD_DT : PS_OPD_DF, e, dt);
        }
    }
    bit1(b);                                           /* reserved_ps */
    return b->pos - at;
}

/* Returns the bits the SBR reader must step over; *status = OK or the error that cleared ps->start. */
static int read_ps(Bits *host, HeaacPsState *ps, int bits_left_in_ext, int *status)
{
    Bits gb = *host, *b = &gb;
    const int at = b->pos;
    *status = HEAAC_PARSE_ERR_DATA;

    const unsigned header = bit1(b);
    if (header) {
        ps->enable_iid = (uint8_t)bit1(b);
        if (ps->enable_iid) {
            const int iid_mode = (int)bits(b, 3);
            if (iid_mode > 5) goto err;
            ps->nr_iid_par    = (uint8_t)k_nr_iidicc_par[iid_mode];
            ps->iid_quant     = iid_mode > 2;
            ps->nr_ipdopd_par = (uint8_t)k_nr_ipdopd_par[iid_mode];
        }
        ps->enable_icc = (uint8_t)bit1(b);
        if (ps->enable_icc) {
            const int icc_mode = (int)bits(b, 3);
            if (icc_mode > 5) goto err;                /* (the reference stores the reserved value first) */
            ps->icc_mode = (uint8_t)icc_mode;
            ps->nr_icc_par = (uint8_t)k_nr_iidicc_par[icc_mode];
        }
        ps->enable_ext = (uint8_t)bit1(b);
    }
    ps->frame_class = (uint8_t)bit1(b);
    ps->num_env_old = ps->num_env;
    ps->num_env     = (uint8_t)k_num_env_tab[ps->frame_class][bits(b, 2)];

    ps->border_position[0] = -1;
    if (ps->frame_class) {
        for (int e = 1; e <= ps->num_env; e++) ps->border_position[e] = (int8_t)bits(b, 5);
    } else {
        const int shift = ps->num_env == 4 ? 2 : ps->num_env == 2 ? 1 : 0;   /* ff_log2_tab[num_env], num_env in {0,1,2,4} */
        for (int e = 1; e <= ps->num_env; e++) ps->border_position[e] = (int8_t)((e * 32 >> shift) - 1);
    }
    if (ps->enable_iid) {
        for (int e = 0; e < ps->num_env; e++) {
            const int dt = (int)bit1(b);
            if (read_par(b, ps, ps->iid_par, PAR_IID, ps->nr_iid_par, k_huff_iid[2 * dt + ps->iid_quant], e, dt)) goto err;
        }
    } else {
        memset(ps->iid_par, 0, sizeof(ps->iid_par));
    }
    if (ps->enable_icc) {
        for (int e = 0; e < ps->num_env; e++) {
            const int dt = (int)bit1(b);
            if (read_par(b, ps, ps->icc_par, PAR_ICC, ps->nr_icc_par, dt ? PS_ICC_DT : PS_ICC_DF, e, dt)) goto err;
        }
    } else {
        memset(ps->icc_par, 0, sizeof(ps->icc_par));
    }
    if (ps->enable_ext) {
        int cnt = (int)bits(b, 4);
        if (cnt == 15) cnt += (int)bits(b, 8);
        cnt *= 8;
        while (cnt > 7) {
            const int id = (int)bits(b, 2);
            cnt -= 2 + read_ps_extension(b, ps, id);
        }
        if (cnt < 0) goto err;
        b->pos += cnt;
    }
    /* fix up envelopes (:234-253) */
    if (!ps->num_env || ps->border_position[ps->num_env] < 31) {
        const int source = ps->num_env ? ps->num_env - 1 : ps->num_env_old - 1;
        if (source >= 0 && source != ps->num_env) {
            if (ps->enable_iid) memcpy(ps->iid_par[ps->num_env], ps->iid_par[source], 34);
            if (ps->enable_icc) memcpy(ps->icc_par[ps->num_env], ps->icc_par[source], 34);
            if (ps->enable_ipdopd) {
                memcpy(ps->ipd_par[ps->num_env], ps->ipd_par[source], 34);
                memcpy(ps->opd_par[ps->num_env], ps->opd_par[source], 34);
            }
        }
        ps->num_env++;
        ps->border_position[ps->num_env] = 31;
    }
    ps->is34bands_old = ps->is34bands;
    if (ps->enable_iid || ps->enable_icc)
        ps->is34bands = (ps->enable_iid && ps->nr_iid_par == 34) || (ps->enable_icc && ps->nr_icc_par == 34);
    if (!ps->enable_ipdopd) {
        memset(ps->ipd_par, 0, sizeof(ps->ipd_par));
        memset(ps->opd_par, 0, sizeof(ps->opd_par));
    }
    if (header) ps->start = 1;
    /* stricter than the reference: the borders must ascend, and an envelope borrowed from the previous frame
     * must fit the quantiser of this one (see heaac_parse.h) */
    for (int e = 0; e < ps->num_env; e++)
        if (ps->border_position[e] >= ps->border_position[e + 1]) goto err;
    if (ps->enable_iid)
        for (int e = 0; e < ps->num_env; e++)
            for (int k = 0; k < ps->nr_iid_par; k++)
                if (abs(ps->iid_par[e][k]) > 7 + 8 * ps->iid_quant) goto err;
    /* (a borrowed ICC envelope can hold the value an earlier, refused frame stopped at) */
    if (ps->enable_icc)
        for (int e = 0; e < ps->num_env; e++)
            for (int k = 0; k < ps->nr_icc_par; k++)
                if (ps->icc_par[e][k] < 0 || ps->icc_par[e][k] > 7) goto err;

    {
        const int consumed = b->pos - at;
        if (consumed <= bits_left_in_ext) {
            host->pos += consumed;
            if (b->over) host->over = 1;
            *status = HEAAC_PARSE_OK;
            return consumed;
        }
        *status = HEAAC_PARSE_ERR_OVERREAD;
    }
err:
    ps->start = 0;
    host->pos += bits_left_in_ext;
    return bits_left_in_ext;
}

/* ------------------------------------------------------------------------------------------ */
/* records                                                                                       */
/* ------------------------------------------------------------------------------------------ */
static void emit_ps(const HeaacPsState *s, HeaacPsFrame *p)
{
    memset(p, 0, sizeof(*p));
    p->border_position[0] = -1;
    p->border_position[1] = 31;
    p->num_env = 1;
    p->nr_iid_par = p->nr_icc_par = 20;
    p->nr_ipdopd_par = 11;
    p->is34bands = s->is34bands;
    p->is34bands_old = s->is34bands_old;
    if (!s->start) return;                             /* mono copy: nothing else is read */
    p->start = 1;
    p->num_env = s->num_env;
    p->num_env_old = s->num_env_old;
    p->enable_ipdopd = s->enable_ipdopd;
    p->iid_quant = s->iid_quant;
    p->icc_mode = s->icc_mode;
    /* a parameter set that was never enabled has no count yet; its values are all zero, for which every
     * count maps to the same thing (aacps.c:826-871) */
    p->nr_iid_par = s->nr_iid_par ? s->nr_iid_par : 20;
    p->nr_icc_par = s->nr_icc_par ? s->nr_icc_par : 20;
    p->nr_ipdopd_par = s->nr_ipdopd_par ? s->nr_ipdopd_par : 11;
    memcpy(p->border_position, s->border_position, 6);
    for (int e = 0; e < 5; e++) {
        memcpy(p->iid_par[e], s->iid_par[e], 34);
        memcpy(p->icc_par[e], s->icc_par[e], 34);
        memcpy(p->ipd_par[e], s->ipd_par[e], 17);
        memcpy(p->opd_par[e], s->opd_par[e], 17);
    }
}

static void emit_sbr(const HeaacSbrStream *s, const HeaacSbrHeader *h, int channels, HeaacSbrFrame *f)
{
    memset(f, 0, sizeof(*f));
    f->hdr = (uint16_t)s->hdr;
    f->start = s->start;
    f->reset = s->reset;
    f->kx_old = s->kx[0];
    f->m_old = s->m[0];
    f->bs_coupling = channels == 2 ? s->bs_coupling : 0;
    for (int ch = 0; ch < channels; ch++) {
        const HeaacSbrChanState *c = &s->data[ch];
        HeaacSbrChannel *o = &f->ch[ch];
        o->t_env_num_env_old = c->t_env_num_env_old;
        if (!s->start) continue;                       /* nothing else of the channel is read */
        o->bs_num_env = c->bs_num_env;
        o->bs_num_noise = c->bs_num_noise;
        o->bs_amp_res = c->bs_amp_res;
        o->bs_add_harmonic_flag = c->bs_add_harmonic_flag;
        memcpy(o->bs_freq_res, c->bs_freq_res, 8);
        memcpy(o->t_env, c->t_env, 8);
        memcpy(o->t_q, c->t_q, 3);
        o->e_a[0] = c->e_a[0];
        o->e_a[1] = c->e_a[1];
        memcpy(o->bs_invf_mode, c->bs_invf_mode, 10);
        memcpy(o->bs_add_harmonic, c->bs_add_harmonic, 48);
        /* the bands of each envelope's resolution; what a state row holds beyond them is not data */
        for (int e = 0; e < c->bs_num_env; e++)
            for (int k = 0; k < h->n[c->bs_freq_res[e + 1]]; k++) o->env_facs_q[e][k] = (uint8_t)c->env_facs[e + 1][k];
        for (int e = 0; e < c->bs_num_noise; e++)
            for (int k = 0; k < h->n_q; k++) o->noise_facs_q[e][k] = (uint8_t)c->noise_facs[e + 1][k];
    }
}

void heaac_sbr_no_payload(HeaacSbrStream *st, int channels, HeaacSbrFrame *sbr, HeaacPsFrame *ps)
{
    HeaacSbrStream t = *st;
    t.start = 0;
    t.reset = 0;
    t.kx[0] = t.kx[1];
    t.m[0] = t.m[1];
    emit_sbr(&t, NULL, channels, sbr);
    if (ps) emit_ps(&st->ps, ps);
    st->kx[0] = st->kx[1];
    st->m[0] = st->m[1];
}

/* ------------------------------------------------------------------------------------------ */
/* ff_decode_sbr_extension (aacsbr.c:1044-1090) + read_sbr_data (:982-1020)                      */
/* ------------------------------------------------------------------------------------------ */
int heaac_sbr_parse_payload(HeaacSbrStream *st, HeaacSbrHeaderTable *tab, int sample_rate,
                            const uint8_t *au, int size, int bit, int cnt, int crc,
                            int channels, int allow_ps,
                            HeaacSbrFrame *sbr, HeaacPsFrame *ps, HeaacSbrParseInfo *info)
{
    if (!st || !tab || !au || !sbr || size < 0 || bit < 0 || bit > 8 * size || cnt < 0 ||
        (channels != 1 && channels != 2) || (allow_ps && !ps))
        return HEAAC_PARSE_ERR_ARG;
    pthread_once(&g_once, tables_init);

    Bits gb, *b = &gb;
    bits_init(b, au, size);
    b->pos = bit;
    int ret = HEAAC_PARSE_OK;
    HeaacSbrParseInfo fi = { 0, 0, 0, HEAAC_PARSE_OK };

    st->reset = 0;
    if (crc) bits(b, 10);                              /* bs_sbr_crc_bits: not checked by the reference either */
    st->kx[0] = st->kx[1];
    st->m[0]  = st->m[1];

    if (bit1(b)) {                                     /* bs_header_flag */
        int touched = 0;
        fi.header = 1;
        st->reset = (uint8_t)read_header(st, b, &touched);
        if (st->reset || (touched && st->hdr)) {
            HdrKey k;
            make_key(&k, st, 2 * sample_rate);
            const int idx = table_find_or_add(tab, &k);
            if (idx > 0) {
                st->hdr = (uint32_t)idx;
                st->have_spectrum = 1;
                st->kx[1] = tab->h[idx].kx;
                st->m[1]  = tab->h[idx].m;
            } else {                                   /* sbr_reset failed (:1029-1033) or no room */
                st->start = 0;
                st->have_spectrum = 0;
                ret = idx == -2 ? HEAAC_PARSE_ERR_ARG : HEAAC_PARSE_ERR_DATA;
            }
        }
    }

    if (st->start) {
        const HeaacSbrHeader *h = &tab->h[st->hdr];
        const HeaacSbrChanState keep0 = st->data[0], keep1 = st->data[1];
        const uint8_t keep_coupling = st->bs_coupling;
        const int r = channels == 2 ? read_cpe(st, h, b) : read_sce(st, h, b);
        if (r) {
            st->data[0] = keep0; st->data[1] = keep1; st->bs_coupling = keep_coupling;
            st->start = 0;
            ret = r;
        } else if (bit1(b)) {                          /* bs_extended_data */
            int left = (int)bits(b, 4);
            if (left == 15) left += (int)bits(b, 8);
            left <<= 3;
            while (left > 7) {
                left -= 2;
                const int id = (int)bits(b, 2);
                if (id == EXTENSION_ID_PS && allow_ps) {
                    fi.ps_present = 1;
                    left -= read_ps(b, &st->ps, left, &fi.ps_status);
                } else {                               /* PS signalled absent, or a reserved extension */
                    b->pos += left;
                    left = 0;
                }
            }
            if (left > 0) b->pos += left;
        }
    }
    fi.sbr_bits = b->pos - bit;
    if (b->over && ret == HEAAC_PARSE_OK) ret = HEAAC_PARSE_ERR_OVERREAD;

    emit_sbr(st, &tab->h[st->hdr], channels, sbr);
    if (ps) emit_ps(&st->ps, ps);
    if (info) *info = fi;
    if (ret == HEAAC_PARSE_OK && fi.ps_status) ret = fi.ps_status;
    return ret;
}

/* ------------------------------------------------------------------------------------------ */
/* whole access units                                                                            */
/* ------------------------------------------------------------------------------------------ */
int heaac_heaac_parse_frame(const HeaacAacConfig *cfg, HeaacAacStream *st, HeaacSbrStream *sst,
                            HeaacSbrHeaderTable *tab, const uint8_t *au, int size,
                            float *coeffs, HeaacIcs *ics, HeaacToolsFrame *tools,
                            HeaacSbrFrame *sbr, HeaacPsFrame *ps, HeaacAacFrameInfo *info)
{
    HeaacAacFrameInfo fi;
    memset(&fi, 0, sizeof(fi));
    if (info) *info = fi;
    if (!cfg || !sst || !tab || !sbr) return HEAAC_PARSE_ERR_ARG;
    const int r = heaac_aac_parse_frame(cfg, st, au, size, coeffs, ics, tools, &fi);
    if (r) return r;                                   /* info->channels = 0: the core element failed */
    if (info) *info = fi;
    const int allow_ps = cfg->ps != 0 && fi.channels == 1 && ps != NULL;
    if (fi.sbr_payload_bit < 0 || cfg->sbr == 0) {
        heaac_sbr_no_payload(sst, fi.channels, sbr, ps);
        return HEAAC_PARSE_NO_SBR;
    }
    return heaac_sbr_parse_payload(sst, tab, cfg->sample_rate, au, size, fi.sbr_payload_bit, fi.sbr_payload_bytes,
                                   fi.sbr_crc, fi.channels, allow_ps, sbr, ps, NULL);
}

typedef struct {
    const HeaacAacConfig *cfg; HeaacAacStream *st; HeaacSbrStream *sst; HeaacSbrHeaderTable *tab;
    const uint8_t *const *au; const int *size;
    float *coeffs; HeaacIcs *ics; HeaacToolsFrame *tools; HeaacSbrFrame *sbr; HeaacPsFrame *ps;
    HeaacAacFrameInfo *info; int *status;
    size_t lo, hi; int failed;
} Job;

static void *job_run(void *p)
{
    Job *j = (Job *)p;
    for (size_t i = j->lo; i < j->hi; i++) {
        const int r = heaac_heaac_parse_frame(j->cfg, &j->st[i], &j->sst[i], j->tab, j->au[i], j->size[i],
                                              j->coeffs + i * 2048, j->ics + 2 * i, &j->tools[i],
                                              &j->sbr[i], j->ps ? &j->ps[i] : NULL, j->info ? &j->info[i] : NULL);
        if (j->status) j->status[i] = r;
        j->failed += r < 0;
    }
    return NULL;
}

int heaac_heaac_parse_batch(const HeaacAacConfig *cfg, HeaacAacStream *st, HeaacSbrStream *sst,
                            HeaacSbrHeaderTable *tab,
                            const uint8_t *const *au, const int *size, size_t n,
                            float *coeffs, HeaacIcs *ics, HeaacToolsFrame *tools,
                            HeaacSbrFrame *sbr, HeaacPsFrame *ps,
                            HeaacAacFrameInfo *info, int *status, int threads)
{
    if (!cfg || !st || !sst || !tab || !au || !size || !coeffs || !ics || !tools || !sbr) return HEAAC_PARSE_ERR_ARG;
    if (threads <= 0) threads = (int)sysconf(_SC_NPROCESSORS_ONLN);
    if (threads < 1) threads = 1;
    if ((size_t)threads > n) threads = n ? (int)n : 1;
    if (threads > 256) threads = 256;
    Job job[256];
    pthread_t tid[256];
    int started[256];
    for (int t = 0; t < threads; t++) {
        job[t] = (Job){ cfg, st, sst, tab, au, size, coeffs, ics, tools, sbr, ps, info, status,
                        n * (size_t)t / (size_t)threads, n * (size_t)(t + 1) / (size_t)threads, 0 };
        started[t] = t > 0 && pthread_create(&tid[t], NULL, job_run, &job[t]) == 0;
    }
    job_run(&job[0]);
    int failed = job[0].failed;
    for (int t = 1; t < threads; t++) {
        if (started[t]) pthread_join(tid[t], NULL);
        else job_run(&job[t]);
        failed += job[t].failed;
    }
    return failed;
}
