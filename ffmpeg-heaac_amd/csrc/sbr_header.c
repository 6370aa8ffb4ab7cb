/* sbr_header.c -- host side: SBR header -> frequency band tables.
 *
 * Product code (C99, no GPU).  Produces the HeaacSbrHeader records the kernels
 * index; replaces sbr_reset() = sbr_make_f_master() + sbr_make_f_derived()
 * (libavcodec/aacsbr.c:1019-1034, :332-490, :544-593), which in turn call
 * make_bands (:296-313), sbr_hf_calc_npatches (:493-541) and
 * sbr_make_f_tablelim (:146-205).  ISO/IEC 14496-3 4.6.18.3.2.
 *
 * Band tables are tiny integer vectors; they are built in `int` arrays with a
 * small insertion sort instead of the reference's qsort-on-int16 plumbing.
 */
#include <math.h>
#include <string.h>
#include "heaac_dsp.h"

#define MAXB 64

typedef struct FreqTables {
    int fs;                     /* SBR (output) sample rate          */
    int k0, k1, k2;             /* start / region border / stop      */
    int kx, m;                  /* first SBR band, number of bands   */
    int n_master, n_lo, n_hi, n_q, n_lim, n_patch;
    int master[MAXB], lo[MAXB], hi[MAXB], noise[8], lim[MAXB];
    int patch_len[8], patch_src[8];
} FreqTables;

static void sort_ints(int *v, int n)
{
    for (int i = 1; i < n; i++) {
        int x = v[i], j = i - 1;
        while (j >= 0 && v[j] > x) {
            v[j + 1] = v[j];
            j--;
        }
        v[j + 1] = x;
    }
}

/* Geometric band widths between start and stop (aacsbr.c:296-313). */
static void geometric_widths(int *width, int start, int stop, int count)
{
    float ratio = powf((float)stop / start, 1.0f / count);
    float edge  = start;
    int   prev  = start;
    for (int i = 0; i < count - 1; i++) {
        edge *= ratio;
        int cur = (int)lrintf(edge);
        width[i] = cur - prev;
        prev = cur;
    }
    width[count - 1] = stop - prev;
}

static int offset_row(int fs)
{
    switch (fs) {
    case 16000: return 0;
    case 22050: return 1;
    case 24000: return 2;
    case 32000: return 3;
    case 44100: case 48000: case 64000: return 4;
    case 88200: case 96000: case 128000: case 176400: case 192000: return 5;
    }
    return -1;
}

/* Master table (aacsbr.c:332-490). */
static int build_master(FreqTables *t, int start_freq, int stop_freq, int xover,
                        int freq_scale, int alter_scale)
{
    /* ISO/IEC 14496-3 Table 4.A.?? start-band offsets, by SBR rate class */
    static const signed char start_offset[6][16] = {
        {-8, -7, -6, -5, -4, -3, -2, -1,  0,  1,  2,  3,  4,  5,  6,  7},
        {-5, -4, -3, -2, -1,  0,  1,  2,  3,  4,  5,  6,  7,  9, 11, 13},
        {-5, -3, -2, -1,  0,  1,  2,  3,  4,  5,  6,  7,  9, 11, 13, 16},
        {-6, -4, -2, -1,  0,  1,  2,  3,  4,  5,  6,  7,  9, 11, 13, 16},
        {-4, -2, -1,  0,  1,  2,  3,  4,  5,  6,  7,  9, 11, 13, 16, 20},
        {-2, -1,  0,  1,  2,  3,  4,  5,  6,  7,  9, 11, 13, 16, 20, 24},
    };
    const int row = offset_row(t->fs);
    if (row < 0 || start_freq < 0 || start_freq > 15 || stop_freq < 0 || stop_freq > 15)
        return -1;

    const unsigned base = t->fs < 32000 ? 3000 : t->fs < 64000 ? 4000 : 5000;
    const int start_min = (int)(((base << 7) + (unsigned)(t->fs >> 1)) / (unsigned)t->fs);
    const int stop_min  = (int)(((base << 8) + (unsigned)(t->fs >> 1)) / (unsigned)t->fs);

    t->k0 = start_min + start_offset[row][start_freq];
    if (t->k0 <= 0)
        return -1;

    if (stop_freq < 14) {
        int w[13];
        geometric_widths(w, stop_min, 64, 13);
        sort_ints(w, 13);
        t->k2 = stop_min;
        for (int i = 0; i < stop_freq; i++)
            t->k2 += w[i];
    } else {
        t->k2 = (stop_freq == 14 ? 2 : 3) * t->k0;
    }
    if (t->k2 > 64)
        t->k2 = 64;

    const int max_bands = t->fs <= 32000 ? 48 : t->fs == 44100 ? 35 : 32;
    if (t->k2 - t->k0 > max_bands || t->k2 <= t->k0)
        return -1;

    if (freq_scale == 0) {
        /* linear spacing */
        const int dk = alter_scale + 1;
        const int n = ((t->k2 - t->k0 + (dk & 2)) >> dk) << 1;
        if (n <= 0 || xover >= n || n >= MAXB)
            return -1;
        int width[MAXB];
        for (int i = 1; i <= n; i++)
            width[i] = dk;
        const int rest = t->k2 - t->k0 - n * dk;
        if (rest < 0) {
            width[1]--;
            if (rest < -1)
                width[2]--;
        } else if (rest > 0) {
            width[n]++;
        }
        t->master[0] = t->k0;
        for (int i = 1; i <= n; i++)
            t->master[i] = t->master[i - 1] + width[i];
        t->n_master = n;
        return 0;
    }

    /* logarithmic spacing, one or two regions */
    const int bands_per_half_octave = 7 - freq_scale;
    const int two = 49 * t->k2 > 110 * t->k0;
    t->k1 = two ? 2 * t->k0 : t->k2;

    const int n0 = (int)lrintf(bands_per_half_octave * log2f(t->k1 / (float)t->k0)) * 2;
    if (n0 <= 0 || n0 >= MAXB)
        return -1;
    int w0[MAXB];
    geometric_widths(w0, t->k0, t->k1, n0);
    sort_ints(w0, n0);
    const int w0_max = w0[n0 - 1];
    t->master[0] = t->k0;
    for (int i = 0; i < n0; i++) {
        if (w0[i] <= 0)
            return -1;
        t->master[i + 1] = t->master[i] + w0[i];
    }
    t->n_master = n0;

    if (two) {
        const float invwarp = alter_scale ? 0.76923076923076923077f : 1.0f;
        const int n1 = (int)lrintf(bands_per_half_octave * invwarp *
                                   log2f(t->k2 / (float)t->k1)) * 2;
        if (n1 <= 0 || n0 + n1 >= MAXB)
            return -1;
        int w1[MAXB];
        geometric_widths(w1, t->k1, t->k2, n1);
        int w1_min = w1[0];
        for (int i = 1; i < n1; i++)
            if (w1[i] < w1_min)
                w1_min = w1[i];
        if (w1_min < w0_max) {
            /* widen the narrowest upper band at the cost of the widest */
            sort_ints(w1, n1);
            int a = w0_max - w1[0], b = (w1[n1 - 1] - w1[0]) >> 1;
            int change = a > b ? b : a;
            w1[0]      += change;
            w1[n1 - 1] -= change;
        }
        sort_ints(w1, n1);
        for (int i = 0; i < n1; i++) {
            if (w1[i] <= 0)
                return -1;
            t->master[n0 + i + 1] = t->master[n0 + i] + w1[i];
        }
        t->n_master = n0 + n1;
    }
    if (xover >= t->n_master)
        return -1;
    return 0;
}

/* Patch construction (aacsbr.c:493-541, 14496-3 fig. 4.46). */
static int build_patches(FreqTables *t)
{
    const int goal = (int)(((1000u << 11) + (unsigned)(t->fs >> 1)) / (unsigned)t->fs);
    int msb = t->k0, usb = t->kx, sb = 0, k;

    t->n_patch = 0;
    if (goal < t->kx + t->m) {
        k = 0;
        while (t->master[k] < goal)
            k++;
    } else {
        k = t->n_master;
    }

    do {
        int odd = 0, i = k;
        /* highest master border not above k0 - 1 + msb - odd, starting at k */
        do {
            sb  = t->master[i];
            odd = (sb + t->k0) & 1;
            i--;
        } while (sb > t->k0 - 1 + msb - odd);

        if (t->n_patch > 5)
            return -1;
        int len = sb - usb;
        if (len < 0)
            len = 0;
        t->patch_len[t->n_patch] = len;
        t->patch_src[t->n_patch] = t->k0 - odd - len;
        if (len > 0) {
            usb = msb = sb;
            t->n_patch++;
        } else {
            msb = t->kx;
        }
        if (t->master[k] - sb < 3)
            k = t->n_master;
    } while (sb != t->kx + t->m);

    if (t->n_patch > 1 && t->patch_len[t->n_patch - 1] < 3)
        t->n_patch--;
    /* ISO/IEC 14496-3 allows at most 5 patches.  The reference lets 6 through
     * (aacsbr.c:516-519) and then writes past f_tablelim[29]; undefined there,
     * rejected here. */
    if (t->n_patch > 5)
        return -1;
    return 0;
}

static int is_patch_border(const int *borders, int n, int v)
{
    for (int i = 0; i <= n; i++)
        if (borders[i] == v)
            return 1;
    return 0;
}

/* Limiter table (aacsbr.c:146-205). */
static void build_limiter(FreqTables *t, int limiter_bands)
{
    if (limiter_bands <= 0) {
        t->lim[0] = t->lo[0];
        t->lim[1] = t->lo[t->n_lo];
        t->n_lim = 1;
        return;
    }
    static const float warp[3] = { 1.32715174233856803909f,    /* 2^(0.49/1.2) */
                                   1.18509277094158210129f,    /* 2^(0.49/2)   */
                                   1.11987160404675912501f };  /* 2^(0.49/3)   */
    const float limit = warp[limiter_bands - 1];
    int borders[8];
    borders[0] = t->kx;
    for (int i = 1; i <= t->n_patch; i++)
        borders[i] = borders[i - 1] + t->patch_len[i - 1];

    int cnt = 0;
    for (int i = 0; i <= t->n_lo; i++)
        t->lim[cnt++] = t->lo[i];
    for (int i = 1; i < t->n_patch; i++)
        t->lim[cnt++] = borders[i];
    sort_ints(t->lim, cnt);         /* cnt == n_lo + n_patch */

    /* thin the merged list: keep a border only if it is far enough (in
     * octaves) from the last kept one, preferring patch borders */
    int n_lim = t->n_lo + t->n_patch - 1;
    int out = 0, in = 1;
    while (out < n_lim) {
        if (t->lim[in] >= t->lim[out] * limit) {
            t->lim[++out] = t->lim[in++];
        } else if (t->lim[in] == t->lim[out] ||
                   !is_patch_border(borders, t->n_patch, t->lim[in])) {
            in++;
            n_lim--;
        } else if (!is_patch_border(borders, t->n_patch, t->lim[out])) {
            t->lim[out] = t->lim[in++];
            n_lim--;
        } else {
            t->lim[++out] = t->lim[in++];
        }
    }
    t->n_lim = n_lim;
}

/* Derived tables (aacsbr.c:544-593). */
static int build_derived(FreqTables *t, int xover, int noise_bands, int limiter_bands)
{
    t->n_hi = t->n_master - xover;
    t->n_lo = (t->n_hi + 1) >> 1;
    for (int i = 0; i <= t->n_hi; i++)
        t->hi[i] = t->master[xover + i];
    t->kx = t->hi[0];
    t->m  = t->hi[t->n_hi] - t->hi[0];
    if (t->kx + t->m > 64 || t->kx > 32)
        return -1;

    const int odd = t->n_hi & 1;
    t->lo[0] = t->hi[0];
    for (int i = 1; i <= t->n_lo; i++)
        t->lo[i] = t->hi[2 * i - odd];

    long nq = lrintf(noise_bands * log2f(t->k2 / (float)t->kx));
    t->n_q = nq < 1 ? 1 : (int)nq;
    if (t->n_q > 5)
        return -1;
    t->noise[0] = t->lo[0];
    for (int i = 1, idx = 0; i <= t->n_q; i++) {
        idx += (t->n_lo - idx) / (t->n_q + 1 - i);
        t->noise[i] = t->lo[idx];
    }

    if (build_patches(t) < 0)
        return -1;
    build_limiter(t, limiter_bands);
    return 0;
}

int heaac_sbr_make_header(HeaacSbrHeader *h, int sample_rate,
                          int bs_start_freq, int bs_stop_freq, int bs_xover_band,
                          int bs_freq_scale, int bs_alter_scale, int bs_noise_bands,
                          int bs_limiter_bands, int bs_limiter_gains,
                          int bs_interpol_freq, int bs_smoothing_mode,
                          int bs_amp_res_header)
{
    FreqTables t;
    if (!h)
        return HEAAC_ERR_ARG;
    memset(&t, 0, sizeof(t));
    memset(h, 0, sizeof(*h));
    t.fs = sample_rate;
    if (bs_xover_band < 0 || bs_freq_scale < 0 || bs_freq_scale > 3 ||
        bs_limiter_bands < 0 || bs_limiter_bands > 3 ||
        bs_limiter_gains < 0 || bs_limiter_gains > 3 ||
        bs_noise_bands < 0 || bs_noise_bands > 3)
        return HEAAC_ERR_ARG;
    if (build_master(&t, bs_start_freq, bs_stop_freq, bs_xover_band,
                     bs_freq_scale, !!bs_alter_scale) < 0)
        return HEAAC_ERR_ARG;
    if (build_derived(&t, bs_xover_band, bs_noise_bands, bs_limiter_bands) < 0)
        return HEAAC_ERR_ARG;

    h->k0 = (uint8_t)t.k0;  h->k2 = (uint8_t)t.k2;
    h->kx = (uint8_t)t.kx;  h->m  = (uint8_t)t.m;
    h->n[0] = (uint8_t)t.n_lo;  h->n[1] = (uint8_t)t.n_hi;
    h->n_q = (uint8_t)t.n_q;    h->n_lim = (uint8_t)t.n_lim;
    h->n_master = (uint8_t)t.n_master;
    h->num_patches = (uint8_t)t.n_patch;
    h->bs_limiter_gains  = (uint8_t)bs_limiter_gains;
    h->bs_interpol_freq  = (uint8_t)!!bs_interpol_freq;
    h->bs_smoothing_mode = (uint8_t)!!bs_smoothing_mode;
    h->bs_amp_res_header = (uint8_t)!!bs_amp_res_header;
    for (int i = 0; i < 6; i++) {
        h->patch_num_subbands[i]  = (uint8_t)t.patch_len[i];
        h->patch_start_subband[i] = (uint8_t)t.patch_src[i];
    }
    for (int i = 0; i <= t.n_q; i++)  h->f_tablenoise[i] = (uint8_t)t.noise[i];
    for (int i = 0; i <= t.n_lo; i++) h->f_tablelow[i]   = (uint8_t)t.lo[i];
    for (int i = 0; i <= t.n_hi; i++) h->f_tablehigh[i]  = (uint8_t)t.hi[i];
    for (int i = 0; i <= t.n_lim; i++) h->f_tablelim[i]  = (uint8_t)t.lim[i];

    /* per-band lookups for the kernels */
    memset(h->map_hi, 0xff, 6 * 64);
    for (int i = 0; i < t.n_hi; i++) {
        for (int k = t.hi[i]; k < t.hi[i + 1]; k++) h->map_hi[k] = (uint8_t)i;
        h->map_mid[(t.hi[i] + t.hi[i + 1]) >> 1] = (uint8_t)i;
    }
    for (int i = 0; i < t.n_lo; i++)
        for (int k = t.lo[i]; k < t.lo[i + 1]; k++) h->map_lo[k] = (uint8_t)i;
    for (int i = 0; i < t.n_q; i++)
        for (int k = t.noise[i]; k < t.noise[i + 1]; k++) h->map_nq[k] = (uint8_t)i;
    for (int i = 0; i < t.n_lim; i++)
        for (int k = t.lim[i]; k < t.lim[i + 1] && k < 64; k++) h->map_lim[k] = (uint8_t)i;
    for (int j = 0, k = t.kx; j < t.n_patch; j++)
        for (int x = 0; x < t.patch_len[j] && k < 64; x++, k++)
            h->map_src[k] = (uint8_t)(t.patch_src[j] + x);
    return HEAAC_OK;
}
