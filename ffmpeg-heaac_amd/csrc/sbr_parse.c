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
static int g_tables_bad;                              /* a code book did not fit its tree (never with the shipped tables) */

static void tables_init(void)
{
    for (int t = 0; t < 10; t++) {
        g_tables_bad |= tree_build(&g_sbr_tree[t], sbr_huff_code + sbr_huff_first[t], NULL, sbr_huff_bits + sbr_huff_first[t],
                   sbr_huff_first[t + 1] - sbr_huff_first[t]);
        g_tables_bad |= tree_build(&g_ps_tree[t], ps_huff_code + ps_huff_first[t], NULL, ps_huff_bits + ps_huff_first[t],
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
/* time / frequency grid.  VALUES as read_sbr_grid / copy_sbr_grid leave them (aacsbr.c:609-766); */
/* the form is this parser's own: the syntax is read into a description of the two frame ends,   */
/* and the envelope borders, the noise border and the transient envelope are derived from it.    */
/* ------------------------------------------------------------------------------------------ */
/* sbr_grid() transmits, for each END of the frame that the frame class declares variable, an absolute
 * border and up to three relative borders walking inwards from it; a fixed end sits at slot 0 / 16 and
 * has no relative borders (FIXFIX instead spreads 1, 2 or 4 envelopes evenly).  Class bit 1 = the
 * leading end is variable, bit 0 = the trailing end is. */
#define GRID_LEAD_VAR(cls)  (((cls) >> 1) & 1)
#define GRID_TRAIL_VAR(cls) ((cls) & 1)
#define GRID_SLOTS 16                                  /* numTimeSlots (1024-sample frames) */

typedef struct GridSyntax {
    int cls, L;                                        /* frame class, envelopes */
    int lead_abs, trail_abs;                           /* first and last border */
    int n_lead, n_trail;                               /* relative borders at either end */
    int lead_step[3], trail_step[3];                   /* their distances (2 r + 2), walking inwards */
    unsigned pointer;                                  /* bs_pointer; 0 where none is sent */
    uint8_t res[5];                                    /* bs_freq_res of envelope 1..L */
} GridSyntax;

/* bits of bs_pointer for L envelopes: ceil(log2(L + 1)) */
static int grid_pointer_bits(int L) { int n = 0; while ((1 << n) < L + 1) n++; return n; }

static int grid_read_syntax(Bits *b, GridSyntax *g)
{
    memset(g, 0, sizeof(*g));
    g->cls = (int)bits(b, 2);
    g->trail_abs = GRID_SLOTS;
    if (g->cls == FIXFIX) {
        g->L = 1 << bits(b, 2);
        if (g->L > 4) return HEAAC_PARSE_ERR_DATA;     /* "too many SBR envelopes in FIXFIX" */
        const unsigned r = bit1(b);                    /* one resolution bit for all envelopes */
        for (int e = 0; e < g->L; e++) g->res[e] = (uint8_t)r;
        return HEAAC_PARSE_OK;
    }
    /* absolute borders, then the counts, then the relative borders: leading end first */
    if (GRID_LEAD_VAR(g->cls))  g->lead_abs = (int)bits(b, 2);
    if (GRID_TRAIL_VAR(g->cls)) g->trail_abs += (int)bits(b, 2);
    if (GRID_LEAD_VAR(g->cls))  g->n_lead = (int)bits(b, 2);
    if (GRID_TRAIL_VAR(g->cls)) g->n_trail = (int)bits(b, 2);
    g->L = g->n_lead + g->n_trail + 1;
    if (g->L > 5) return HEAAC_PARSE_ERR_DATA;         /* "too many SBR envelopes in VARVAR" */
    for (int i = 0; i < g->n_lead; i++)  g->lead_step[i]  = 2 * (int)bits(b, 2) + 2;
    for (int i = 0; i < g->n_trail; i++) g->trail_step[i] = 2 * (int)bits(b, 2) + 2;
    g->pointer = bits(b, grid_pointer_bits(g->L));
    /* one resolution bit per envelope; a frame with only its trailing end variable sends them last
     * envelope first */
    const int backwards = g->cls == FIXVAR;
    for (int e = 0; e < g->L; e++) g->res[backwards ? g->L - 1 - e : e] = (uint8_t)bit1(b);
    return HEAAC_PARSE_OK;
}

/* Envelope borders t[0..L].  0 = fine; non-zero = two borders meet or cross (the reference's uint8
 * arithmetic wraps a border walked below zero to > 235, which its own monotony check then refuses;
 * meeting borders are refused here as well, see heaac_parse.h). */
static int grid_borders(const GridSyntax *g, int t[6])
{
    const int L = g->L;
    t[0] = g->lead_abs;
    t[L] = g->trail_abs;
    if (g->cls == FIXFIX) {
        for (int e = 1; e < L; e++) t[e] = e * (GRID_SLOTS / L);      /* L in {1, 2, 4} */
    } else {
        for (int i = 0; i < g->n_lead; i++)  t[1 + i] = t[i] + g->lead_step[i];
        for (int i = 0; i < g->n_trail; i++) t[L - 1 - i] = t[L - i] - g->trail_step[i];
    }
    for (int e = 0; e < L; e++)
        if (t[e] >= t[e + 1] || t[e] < 0) return 1;
    return 0;
}

/* Index into t_env[] of the middle noise border (two noise floors only).  ISO/IEC 14496-3 4.6.18.3.3 counts
 * the pointer from the variable end; the reference's expression for a variable trailing end is
 * `bs_num_env - FFMAX(bs_pointer - 1, 1)` on an UNSIGNED bs_pointer (aacsbr.c:613, 729): with bs_pointer = 0
 * the subtraction wraps and the index comes out as L + 1, one past the last border -- t_env[] keeps what an
 * earlier frame with more envelopes left there (0 in a new stream).  The reference is the contract, so the
 * same entry is taken here (ISO would give L - 1); tests/test_sbr_parse.py pins the case. */
static int grid_noise_border_index(const GridSyntax *g)
{
    const int L = g->L, p = (int)g->pointer;
    if (g->cls == FIXFIX) return L >> 1;
    if (GRID_TRAIL_VAR(g->cls)) return p == 0 ? L + 1 : L - (p > 2 ? p - 1 : 1);
    return p == 0 ? 1 : p == 1 ? L - 1 : p - 1;        /* VARFIX: counted from the leading end */
}

/* l_A: the envelope that starts at the transient, or -1 */
static int grid_transient_envelope(const GridSyntax *g)
{
    const int p = (int)g->pointer;
    if (GRID_TRAIL_VAR(g->cls)) return p ? g->L + 1 - p : -1;
    if (g->cls == VARFIX) return p > 1 ? p - 1 : -1;
    return -1;
}

/* What a channel keeps of the PREVIOUS frame's grid when a new one arrives: the resolution and the end of
 * its last envelope, and whether its transient envelope was its last (l_APrev). */
static void grid_carry(HeaacSbrChanState *c)
{
    const int L_old = c->bs_num_env;
    c->bs_freq_res[0] = c->bs_freq_res[L_old];
    c->t_env_num_env_old = c->t_env[L_old];
    c->e_a[0] = (int8_t)-(c->e_a[1] != L_old);
}

static int read_grid(const HeaacSbrStream *s, Bits *b, HeaacSbrChanState *c)
{
    GridSyntax g;
    int t[6];
    /* the previous frame's values move first, as in the reference, also when this grid is refused */
    c->bs_freq_res[0] = c->bs_freq_res[c->bs_num_env];
    c->t_env_num_env_old = c->t_env[c->bs_num_env];
    const int L_old = c->bs_num_env;
    const int rc = grid_read_syntax(b, &g);
    if (rc) return rc;                                 /* (the caller rolls a refused element's channel state back) */
    c->bs_frame_class = (uint8_t)g.cls;
    c->bs_amp_res = (g.cls == FIXFIX && g.L == 1) ? 0 : s->bs_amp_res_header;
    c->bs_num_env = (uint8_t)g.L;
    const int crossed = grid_borders(&g, t);
    for (int e = 0; e <= g.L; e++) c->t_env[e] = (uint8_t)t[e];
    for (int e = 0; e < g.L; e++) c->bs_freq_res[1 + e] = g.res[e];
    if (g.pointer > (unsigned)g.L + 1 || crossed) return HEAAC_PARSE_ERR_DATA;

    c->bs_num_noise = (uint8_t)(g.L > 1 ? 2 : 1);
    c->t_q[0] = c->t_env[0];
    c->t_q[c->bs_num_noise] = c->t_env[g.L];
    if (c->bs_num_noise > 1) c->t_q[1] = c->t_env[grid_noise_border_index(&g)];
    c->e_a[0] = (int8_t)-(c->e_a[1] != L_old);
    c->e_a[1] = (int8_t)grid_transient_envelope(&g);
    return HEAAC_PARSE_OK;
}

/* The second channel of a coupled pair takes the first one's grid (copy_sbr_grid): its own carries, then
 * every transmitted grid field of the partner. */
static void copy_grid(HeaacSbrChanState *dst, const HeaacSbrChanState *src)
{
    grid_carry(dst);
    dst->bs_frame_class = src->bs_frame_class;
    dst->bs_num_env = src->bs_num_env;
    dst->bs_num_noise = src->bs_num_noise;
    dst->bs_amp_res = src->bs_amp_res;
    dst->e_a[1] = src->e_a[1];
    for (int i = 0; i < 8; i++) dst->t_env[i] = src->t_env[i];
    for (int i = 1; i < 8; i++) dst->bs_freq_res[i] = src->bs_freq_res[i];
    for (int i = 0; i < 3; i++) dst->t_q[i] = src->t_q[i];
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
/* Parametric Stereo payload.  VALUES as ff_ps_read_data leaves them (aacps.c:84-279); the form  */
/* is this parser's own: ps_data() is read as a mode header, a time grid and then one block per   */
/* parameter KIND, each kind described by a record of where it lives and how it is coded.        */
/* ------------------------------------------------------------------------------------------ */
/* resolution mode 0..5 -> parameters per envelope (ISO/IEC 14496-3 Table 8.4x): modes 3..5 repeat 0..2 with
 * the fine IID quantiser */
static int ps_mode_bands(int mode)  { static const int8_t n[3] = { 10, 20, 34 }; return n[mode % 3]; }
static int ps_mode_phases(int mode) { static const int8_t n[3] = { 5, 11, 17 };  return n[mode % 3]; }

/* One kind of stereo parameter of one frame. */
typedef struct PsKind {
    int8_t (*par)[34];            /* [envelope][band] */
    int count;                    /* values per envelope */
    int book[2];                  /* code book by direction: [0] along frequency, [1] along time */
    int bias[2];                  /* what that code book adds to a difference */
    int modulo;                   /* phases wrap modulo 8; 0 = plain integers */
    int lo, hi;                   /* legal range of an accumulated value (ignored for phases) */
} PsKind;

static PsKind ps_kind_iid(HeaacPsState *ps)
{
    const int fine = ps->iid_quant;
    const int df = fine ? PS_IID_DF1 : PS_IID_DF0, dt = fine ? PS_IID_DT1 : PS_IID_DT0;
    const int top = fine ? 15 : 7;
    const PsKind k = { ps->iid_par, ps->nr_iid_par, { df, dt }, { ps_huff_offset[df], ps_huff_offset[dt] }, 0, -top, top };
    return k;
}
static PsKind ps_kind_icc(HeaacPsState *ps)
{
    const PsKind k = { ps->icc_par, ps->nr_icc_par, { PS_ICC_DF, PS_ICC_DT },
                       { ps_huff_offset[PS_ICC_DF], ps_huff_offset[PS_ICC_DT] }, 0, 0, 7 };
    return k;
}
static PsKind ps_kind_phase(HeaacPsState *ps, int opd)
{
    const PsKind k = { opd ? ps->opd_par : ps->ipd_par, ps->nr_ipdopd_par,
                       { opd ? PS_OPD_DF : PS_IPD_DF, opd ? PS_OPD_DT : PS_IPD_DT }, { 0, 0 }, 8, 0, 0 };
    return k;
}

/* One envelope of one kind: a direction bit, then `count` code words.  Along frequency a value continues
 * from its lower neighbour (from 0 for the first), along time from the same band of the previous envelope --
 * for envelope 0 the last one of the previous frame. */
static int ps_read_envelope(Bits *b, const HeaacPsState *ps, const PsKind *k, int e)
{
    const int along_time = (int)bit1(b);
    const Tree *book = &g_ps_tree[k->book[along_time]];
    const int8_t *before = NULL;
    if (along_time) {
        const int src = e > 0 ? e - 1 : ps->num_env_old > 0 ? ps->num_env_old - 1 : 0;
        before = k->par[src];
    }
    int run = 0;
    for (int band = 0; band < k->count; band++) {
        const int sym = tree_read(book, b);
        if (sym < 0) return HEAAC_PARSE_ERR_DATA;
        run = (before ? before[band] : run) + sym - k->bias[along_time];
        if (k->modulo) run &= k->modulo - 1;
        else if (run < k->lo || run > k->hi) { k->par[e][band] = (int8_t)run; return HEAAC_PARSE_ERR_DATA; }
        k->par[e][band] = (int8_t)run;
    }
    return HEAAC_PARSE_OK;
}

/* All envelopes of a kind that is switched on; a kind that is off reads as zeros. */
static int ps_read_kind(Bits *b, HeaacPsState *ps, int enabled, PsKind k)
{
    if (!enabled) { memset(k.par, 0, 5 * 34); return HEAAC_PARSE_OK; }
    for (int e = 0; e < ps->num_env; e++)
        if (ps_read_envelope(b, ps, &k, e)) return HEAAC_PARSE_ERR_DATA;
    return HEAAC_PARSE_OK;
}

/* enable_ps_header: which kinds are on and at which resolution.  0, or non-zero for a reserved mode. */
static int ps_read_modes(Bits *b, HeaacPsState *ps)
{
    if ((ps->enable_iid = (uint8_t)bit1(b))) {
        const int mode = (int)bits(b, 3);
        if (mode > 5) return 1;
        ps->nr_iid_par = (uint8_t)ps_mode_bands(mode);
        ps->nr_ipdopd_par = (uint8_t)ps_mode_phases(mode);
        ps->iid_quant = mode >= 3;
    }
    if ((ps->enable_icc = (uint8_t)bit1(b))) {
        const int mode = (int)bits(b, 3);
        if (mode > 5) return 1;                        /* (the reference stores the reserved value before refusing it) */
        ps->icc_mode = (uint8_t)mode;
        ps->nr_icc_par = (uint8_t)ps_mode_bands(mode);
    }
    ps->enable_ext = (uint8_t)bit1(b);
    return 0;
}

/* Envelope count and borders: class 0 = 0, 1, 2 or 4 envelopes cutting the 32 slots evenly, class 1 = 1..4
 * envelopes with transmitted borders.  border_position[0] = -1. */
static void ps_read_time_grid(Bits *b, HeaacPsState *ps)
{
    ps->frame_class = (uint8_t)bit1(b);
    const int code = (int)bits(b, 2);
    ps->num_env_old = ps->num_env;
    ps->num_env = (uint8_t)(ps->frame_class ? code + 1 : code == 3 ? 4 : code);
    ps->border_position[0] = -1;
    for (int e = 1; e <= ps->num_env; e++)
        ps->border_position[e] = (int8_t)(ps->frame_class ? (int)bits(b, 5) : e * 32 / ps->num_env - 1);
}

/* ps_extension(): id 0 carries the phase parameters (IPD and OPD interleaved per envelope) and a
 * reserved bit; other ids carry nothing this decoder reads.  Returns the bits consumed. */
static int ps_read_extension(Bits *b, HeaacPsState *ps, int id)
{
    const int at = b->pos;
    if (id != 0) return 0;
    if ((ps->enable_ipdopd = (uint8_t)bit1(b))) {
        const PsKind ipd = ps_kind_phase(ps, 0), opd = ps_kind_phase(ps, 1);
        for (int e = 0; e < ps->num_env; e++) {
            ps_read_envelope(b, ps, &ipd, e);
            ps_read_envelope(b, ps, &opd, e);
        }
    }
    bit1(b);                                           /* reserved_ps */
    return b->pos - at;
}

/* The extension container: a byte count (escaped at 15), then extensions while at least one byte
 * remains; what is left is padding.  Non-zero if the extensions ran past the count. */
static int ps_read_extensions(Bits *b, HeaacPsState *ps)
{
    int left = (int)bits(b, 4);
    if (left == 15) left += (int)bits(b, 8);
    left *= 8;
    while (left > 7) {
        const int id = (int)bits(b, 2);
        left -= 2 + ps_read_extension(b, ps, id);
    }
    if (left < 0) return 1;
    b->pos += left;
    return 0;
}

/* The last envelope must reach the end of the frame (slot 31): if it does not -- or no envelope was
 * sent -- one more is appended that repeats the last parameters known (this frame's last envelope, or
 * the previous frame's).  aacps.c:234-253. */
static void ps_close_time_grid(HeaacPsState *ps)
{
    const int n = ps->num_env;
    if (n && ps->border_position[n] >= 31) return;
    const int from = n ? n - 1 : (int)ps->num_env_old - 1;
    if (from >= 0 && from != n) {
        int8_t (*const sets[4])[34] = { ps->iid_par, ps->icc_par, ps->ipd_par, ps->opd_par };
        const int on[4] = { ps->enable_iid, ps->enable_icc, ps->enable_ipdopd, ps->enable_ipdopd };
        for (int k = 0; k < 4; k++)
            if (on[k]) memcpy(sets[k][n], sets[k][from], 34);
    }
    ps->num_env = (uint8_t)(n + 1);
    ps->border_position[n + 1] = 31;
}

/* What this parser refuses although the reference goes on (heaac_parse.h): borders that do not ascend, and
 * values outside the quantiser of THIS frame in an envelope borrowed from an earlier one. */
static int ps_frame_is_usable(const HeaacPsState *ps_c)
{
    HeaacPsState *ps = (HeaacPsState *)ps_c;           /* (the kind records are not const-qualified) */
    for (int e = 0; e < ps->num_env; e++)
        if (ps->border_position[e] >= ps->border_position[e + 1]) return 0;
    const PsKind kinds[2] = { ps_kind_iid(ps), ps_kind_icc(ps) };
    const int on[2] = { ps->enable_iid, ps->enable_icc };
    for (int k = 0; k < 2; k++)
        for (int e = 0; on[k] && e < ps->num_env; e++)
            for (int band = 0; band < kinds[k].count; band++)
                if (kinds[k].par[e][band] < kinds[k].lo || kinds[k].par[e][band] > kinds[k].hi) return 0;
    return 1;
}

/* Returns the bits the SBR reader must step over; *status = OK or the error that cleared ps->start. */
static int read_ps(Bits *host, HeaacPsState *ps, int bits_left_in_ext, int *status)
{
    Bits own = *host, *b = &own;                       /* a private cursor: the host only moves on success */
    const int at = b->pos;
    int ok;
    *status = HEAAC_PARSE_ERR_DATA;

    const int has_modes = (int)bit1(b);
    ok = !(has_modes && ps_read_modes(b, ps));
    if (ok) {
        ps_read_time_grid(b, ps);
        ok = ps_read_kind(b, ps, ps->enable_iid, ps_kind_iid(ps)) == HEAAC_PARSE_OK &&
             ps_read_kind(b, ps, ps->enable_icc, ps_kind_icc(ps)) == HEAAC_PARSE_OK;
    }
    if (ok && ps->enable_ext) ok = !ps_read_extensions(b, ps);
    if (ok) {
        ps_close_time_grid(ps);
        /* band layout of this frame: 34 bands as soon as one kind is sent at that resolution; a frame
         * that sends neither keeps the layout it had */
        ps->is34bands_old = ps->is34bands;
        if (ps->enable_iid || ps->enable_icc)
            ps->is34bands = (ps->enable_iid && ps->nr_iid_par == 34) || (ps->enable_icc && ps->nr_icc_par == 34);
        if (!ps->enable_ipdopd) {
            memset(ps->ipd_par, 0, sizeof(ps->ipd_par));
            memset(ps->opd_par, 0, sizeof(ps->opd_par));
        }
        if (has_modes) ps->start = 1;
        ok = ps_frame_is_usable(ps);
    }
    if (ok) {
        const int used = b->pos - at;
        if (used <= bits_left_in_ext) {
            host->pos += used;
            if (b->over) host->over = 1;
            *status = HEAAC_PARSE_OK;
            return used;
        }
        *status = HEAAC_PARSE_ERR_OVERREAD;
    }
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
        (channels != 1 && channels != 2) || (allow_ps & ~(HEAAC_SBR_ALLOW_PS | HEAAC_SBR_MISPLACED)) ||
        ((allow_ps & HEAAC_SBR_ALLOW_PS) && !ps))
        return HEAAC_PARSE_ERR_ARG;
    const int misplaced = allow_ps & HEAAC_SBR_MISPLACED;
    allow_ps &= HEAAC_SBR_ALLOW_PS;
    pthread_once(&g_once, tables_init);
    if (g_tables_bad) return HEAAC_PARSE_ERR_ARG;

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

    if (st->start && misplaced) {
        /* read_sbr_data with the type of a fill / data stream / program config element or an LFE (aacsbr.c:996-1000):
         * "Invalid bitstream - cannot apply SBR to element type %d" */
        st->start = 0;
        ret = HEAAC_PARSE_ERR_DATA;
    } else if (st->start) {
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
    return heaac_heaac_parse_frame_ex(cfg, st, sst, tab, au, size, 2, coeffs, ics, tools, sbr, ps, info);
}

int heaac_heaac_parse_frame_ex(const HeaacAacConfig *cfg, HeaacAacStream *st, HeaacSbrStream *sst,
                               HeaacSbrHeaderTable *tab, const uint8_t *au, int size, int coeff_channels,
                               float *coeffs, HeaacIcs *ics, HeaacToolsFrame *tools,
                               HeaacSbrFrame *sbr, HeaacPsFrame *ps, HeaacAacFrameInfo *info)
{
    HeaacAacFrameInfo fi;
    memset(&fi, 0, sizeof(fi));
    if (info) *info = fi;
    if (!cfg || !sst || !tab || !sbr) return HEAAC_PARSE_ERR_ARG;
    const int r = heaac_aac_parse_frame_ex(cfg, st, au, size, coeff_channels, coeffs, ics, tools, NULL, &fi);
    if (info) *info = fi;                              /* refused: channels = 0 and the HEAAC_REFUSED_* flags */
    if (r) return r;
    const int allow_ps = (cfg->ps != 0 && fi.channels == 1 && ps != NULL ? HEAAC_SBR_ALLOW_PS : 0) |
                         (fi.sbr_misplaced ? HEAAC_SBR_MISPLACED : 0);
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
