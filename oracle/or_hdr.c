/* or_hdr.c -- oracle: SBR header -> frequency band tables
 * (aacsbr.c:146-205 sbr_make_f_tablelim, :296-313 make_bands,
 *  :332-490 sbr_make_f_master, :493-541 sbr_hf_calc_npatches,
 *  :544-593 sbr_make_f_derived).  TEST INFRASTRUCTURE (see oracle.h).
 * Integer logic plus a few float powf/log2f/lrintf calls; kept in the
 * reference's evaluation order.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define FFMIN(a,b) ((a) > (b) ? (b) : (a))
#define FFMAX(a,b) ((a) > (b) ? (a) : (b))

/* aacsbrdata.h:260-267 (ISO/IEC 14496-3 Table 4.A.?? "offset") */
static const int8_t sbr_offset[6][16] = {
    {-8, -7, -6, -5, -4, -3, -2, -1,  0,  1,  2,  3,  4,  5,  6,  7},
    {-5, -4, -3, -2, -1,  0,  1,  2,  3,  4,  5,  6,  7,  9, 11, 13},
    {-5, -3, -2, -1,  0,  1,  2,  3,  4,  5,  6,  7,  9, 11, 13, 16},
    {-6, -4, -2, -1,  0,  1,  2,  3,  4,  5,  6,  7,  9, 11, 13, 16},
    {-4, -2, -1,  0,  1,  2,  3,  4,  5,  6,  7,  9, 11, 13, 16, 20},
    {-2, -1,  0,  1,  2,  3,  4,  5,  6,  7,  9, 11, 13, 16, 20, 24},
};

typedef struct {
    int sample_rate;
    unsigned k[5], kx1, m1, n_master, n[2], n_q, n_lim, num_patches;
    uint16_t f_master[49], f_tablelow[25], f_tablehigh[49], f_tablenoise[6], f_tablelim[29];
    uint8_t patch_num_subbands[6], patch_start_subband[6];
    int bs_limiter_bands, bs_xover_band, bs_noise_bands;
} hdr_ctx;

static int cmp_i16(const void *a, const void *b)
{
    return *(const int16_t *)a - *(const int16_t *)b;
}

static int in_table(const int16_t *table, int last_el, int16_t needle)
{
    int i;
    for (i = 0; i <= last_el; i++)
        if (table[i] == needle)
            return 1;
    return 0;
}

static void make_f_tablelim(hdr_ctx *s)
{
    int k;
    if (s->bs_limiter_bands > 0) {
        static const float bands_warped[3] = { 1.32715174233856803909f,
                                               1.18509277094158210129f,
                                               1.11987160404675912501f };
        const float lim_bands_per_octave_warped = bands_warped[s->bs_limiter_bands - 1];
        int16_t patch_borders[7];
        uint16_t *in = s->f_tablelim + 1, *out = s->f_tablelim;

        patch_borders[0] = s->kx1;
        for (k = 1; k <= (int)s->num_patches; k++)
            patch_borders[k] = patch_borders[k - 1] + s->patch_num_subbands[k - 1];

        memcpy(s->f_tablelim, s->f_tablelow, (s->n[0] + 1) * sizeof(s->f_tablelow[0]));
        if (s->num_patches > 1)
            memcpy(s->f_tablelim + s->n[0] + 1, patch_borders + 1,
                   (s->num_patches - 1) * sizeof(patch_borders[0]));

        qsort(s->f_tablelim, s->num_patches + s->n[0], sizeof(s->f_tablelim[0]), cmp_i16);

        s->n_lim = s->n[0] + s->num_patches - 1;
        while (out < s->f_tablelim + s->n_lim) {
            if (*in >= *out * lim_bands_per_octave_warped) {
                *++out = *in++;
            } else if (*in == *out || !in_table(patch_borders, s->num_patches, *in)) {
                in++;
                s->n_lim--;
            } else if (!in_table(patch_borders, s->num_patches, *out)) {
                *out = *in++;
                s->n_lim--;
            } else {
                *++out = *in++;
            }
        }
    } else {
        s->f_tablelim[0] = s->f_tablelow[0];
        s->f_tablelim[1] = s->f_tablelow[s->n[0]];
        s->n_lim = 1;
    }
}

static void make_bands(int16_t *bands, int start, int stop, int num_bands)
{
    int k, previous, present;
    float base, prod;
    base = powf((float)stop / start, 1.0f / num_bands);
    prod = start;
    previous = start;
    for (k = 0; k < num_bands - 1; k++) {
        prod *= base;
        present  = lrintf(prod);
        bands[k] = present - previous;
        previous = present;
    }
    bands[num_bands - 1] = stop - previous;
}

static int make_f_master(hdr_ctx *s, int bs_start_freq, int bs_stop_freq,
                         int bs_freq_scale, int bs_alter_scale)
{
    unsigned int temp, max_qmf_subbands = 0, start_min, stop_min;
    int k;
    const int8_t *off;
    int16_t stop_dk[13];

    if (s->sample_rate < 32000)      temp = 3000;
    else if (s->sample_rate < 64000) temp = 4000;
    else                             temp = 5000;

    start_min = ((temp << 7) + (s->sample_rate >> 1)) / s->sample_rate;
    stop_min  = ((temp << 8) + (s->sample_rate >> 1)) / s->sample_rate;

    switch (s->sample_rate) {
    case 16000: off = sbr_offset[0]; break;
    case 22050: off = sbr_offset[1]; break;
    case 24000: off = sbr_offset[2]; break;
    case 32000: off = sbr_offset[3]; break;
    case 44100: case 48000: case 64000: off = sbr_offset[4]; break;
    case 88200: case 96000: case 128000: case 176400: case 192000: off = sbr_offset[5]; break;
    default: return -1;
    }

    s->k[0] = start_min + off[bs_start_freq];

    if (bs_stop_freq < 14) {
        s->k[2] = stop_min;
        make_bands(stop_dk, stop_min, 64, 13);
        qsort(stop_dk, 13, sizeof(stop_dk[0]), cmp_i16);
        for (k = 0; k < bs_stop_freq; k++)
            s->k[2] += stop_dk[k];
    } else if (bs_stop_freq == 14) {
        s->k[2] = 2 * s->k[0];
    } else if (bs_stop_freq == 15) {
        s->k[2] = 3 * s->k[0];
    } else
        return -1;
    s->k[2] = FFMIN(64, s->k[2]);

    if (s->sample_rate <= 32000)       max_qmf_subbands = 48;
    else if (s->sample_rate == 44100)  max_qmf_subbands = 35;
    else if (s->sample_rate >= 48000)  max_qmf_subbands = 32;

    if (s->k[2] - s->k[0] > max_qmf_subbands)
        return -1;

    if (!bs_freq_scale) {
        int dk, k2diff;
        dk = bs_alter_scale + 1;
        s->n_master = ((s->k[2] - s->k[0] + (dk & 2)) >> dk) << 1;
        if ((int)s->n_master <= 0 || s->bs_xover_band >= (int)s->n_master)
            return -1;
        for (k = 1; k <= (int)s->n_master; k++)
            s->f_master[k] = dk;
        k2diff = s->k[2] - s->k[0] - s->n_master * dk;
        if (k2diff < 0) {
            s->f_master[1]--;
            s->f_master[2] -= (k2diff < -1);
        } else if (k2diff) {
            s->f_master[s->n_master]++;
        }
        s->f_master[0] = s->k[0];
        for (k = 1; k <= (int)s->n_master; k++)
            s->f_master[k] += s->f_master[k - 1];
    } else {
        int half_bands = 7 - bs_freq_scale;
        int two_regions, num_bands_0;
        int vdk0_max, vdk1_min;
        int16_t vk0[49];

        if (49 * s->k[2] > 110 * s->k[0]) {
            two_regions = 1;
            s->k[1] = 2 * s->k[0];
        } else {
            two_regions = 0;
            s->k[1] = s->k[2];
        }
        num_bands_0 = lrintf(half_bands * log2f(s->k[1] / (float)s->k[0])) * 2;
        if (num_bands_0 <= 0)
            return -1;
        vk0[0] = 0;
        make_bands(vk0 + 1, s->k[0], s->k[1], num_bands_0);
        qsort(vk0 + 1, num_bands_0, sizeof(vk0[1]), cmp_i16);
        vdk0_max = vk0[num_bands_0];
        vk0[0] = s->k[0];
        for (k = 1; k <= num_bands_0; k++) {
            if (vk0[k] <= 0)
                return -1;
            vk0[k] += vk0[k - 1];
        }
        if (two_regions) {
            int16_t vk1[49];
            float invwarp = bs_alter_scale ? 0.76923076923076923077f : 1.0f;
            int num_bands_1 = lrintf(half_bands * invwarp * log2f(s->k[2] / (float)s->k[1])) * 2;
            make_bands(vk1 + 1, s->k[1], s->k[2], num_bands_1);
            vdk1_min = vk1[1];
            for (k = 2; k <= num_bands_1; k++)
                vdk1_min = FFMIN(vk1[k], vdk1_min);
            if (vdk1_min < vdk0_max) {
                int change;
                qsort(vk1 + 1, num_bands_1, sizeof(vk1[1]), cmp_i16);
                change = FFMIN(vdk0_max - vk1[1], (vk1[num_bands_1] - vk1[1]) >> 1);
                vk1[1]           += change;
                vk1[num_bands_1] -= change;
            }
            qsort(vk1 + 1, num_bands_1, sizeof(vk1[1]), cmp_i16);
            vk1[0] = s->k[1];
            for (k = 1; k <= num_bands_1; k++) {
                if (vk1[k] <= 0)
                    return -1;
                vk1[k] += vk1[k - 1];
            }
            s->n_master = num_bands_0 + num_bands_1;
            if ((int)s->n_master <= 0 || s->bs_xover_band >= (int)s->n_master)
                return -1;
            for (k = 0; k <= num_bands_0; k++)
                s->f_master[k] = vk0[k];
            for (k = 1; k <= num_bands_1; k++)
                s->f_master[num_bands_0 + k] = vk1[k];
        } else {
            s->n_master = num_bands_0;
            if ((int)s->n_master <= 0 || s->bs_xover_band >= (int)s->n_master)
                return -1;
            for (k = 0; k <= num_bands_0; k++)
                s->f_master[k] = vk0[k];
        }
    }
    return 0;
}

static int hf_calc_npatches(hdr_ctx *s)
{
    int i, k, sb = 0;
    int msb = s->k[0];
    int usb = s->kx1;
    int goal_sb = ((1000 << 11) + (s->sample_rate >> 1)) / s->sample_rate;

    s->num_patches = 0;
    if (goal_sb < (int)(s->kx1 + s->m1)) {
        for (k = 0; s->f_master[k] < goal_sb; k++) ;
    } else
        k = s->n_master;

    do {
        int odd = 0;
        for (i = k; i == k || sb > ((int)s->k[0] - 1 + msb - odd); i--) {
            sb = s->f_master[i];
            odd = (sb + s->k[0]) & 1;
        }
        if (s->num_patches > 5)
            return -1;
        s->patch_num_subbands[s->num_patches]  = FFMAX(sb - usb, 0);
        s->patch_start_subband[s->num_patches] = s->k[0] - odd - s->patch_num_subbands[s->num_patches];
        if (s->patch_num_subbands[s->num_patches] > 0) {
            usb = sb;
            msb = sb;
            s->num_patches++;
        } else
            msb = s->kx1;
        if (s->f_master[k] - sb < 3)
            k = s->n_master;
    } while (sb != (int)(s->kx1 + s->m1));

    if (s->patch_num_subbands[s->num_patches - 1] < 3 && s->num_patches > 1)
        s->num_patches--;
    /* The reference tolerates a final count of 6 (comment at aacsbr.c:516-519)
     * but then overruns f_tablelim[29] in sbr_make_f_tablelim; that case is
     * undefined there and rejected here and in the product. */
    if (s->num_patches > 5)
        return -1;
    return 0;
}

static int make_f_derived(hdr_ctx *s)
{
    int k, temp;
    s->n[1] = s->n_master - s->bs_xover_band;
    s->n[0] = (s->n[1] + 1) >> 1;
    memcpy(s->f_tablehigh, &s->f_master[s->bs_xover_band], (s->n[1] + 1) * sizeof(s->f_master[0]));
    s->m1  = s->f_tablehigh[s->n[1]] - s->f_tablehigh[0];
    s->kx1 = s->f_tablehigh[0];
    if (s->kx1 + s->m1 > 64)
        return -1;
    if (s->kx1 > 32)
        return -1;
    s->f_tablelow[0] = s->f_tablehigh[0];
    temp = s->n[1] & 1;
    for (k = 1; k <= (int)s->n[0]; k++)
        s->f_tablelow[k] = s->f_tablehigh[2 * k - temp];
    s->n_q = FFMAX(1, lrintf(s->bs_noise_bands * log2f(s->k[2] / (float)s->kx1)));
    if (s->n_q > 5)
        return -1;
    s->f_tablenoise[0] = s->f_tablelow[0];
    temp = 0;
    for (k = 1; k <= (int)s->n_q; k++) {
        temp += (s->n[0] - temp) / (s->n_q + 1 - k);
        s->f_tablenoise[k] = s->f_tablelow[temp];
    }
    if (hf_calc_npatches(s) < 0)
        return -1;
    make_f_tablelim(s);
    return 0;
}

int oracle_sbr_make_header(HeaacSbrHeader *h, int sample_rate,
                           int bs_start_freq, int bs_stop_freq, int bs_xover_band,
                           int bs_freq_scale, int bs_alter_scale, int bs_noise_bands,
                           int bs_limiter_bands, int bs_limiter_gains,
                           int bs_interpol_freq, int bs_smoothing_mode,
                           int bs_amp_res_header)
{
    hdr_ctx s;
    int i;
    memset(&s, 0, sizeof(s));
    memset(h, 0, sizeof(*h));
    s.sample_rate = sample_rate;
    s.bs_limiter_bands = bs_limiter_bands;
    s.bs_xover_band = bs_xover_band;
    s.bs_noise_bands = bs_noise_bands;
    if (make_f_master(&s, bs_start_freq, bs_stop_freq, bs_freq_scale, bs_alter_scale) < 0)
        return HEAAC_ERR_ARG;
    if (make_f_derived(&s) < 0)
        return HEAAC_ERR_ARG;
    h->k0 = s.k[0]; h->k2 = s.k[2]; h->kx = s.kx1; h->m = s.m1;
    h->n[0] = s.n[0]; h->n[1] = s.n[1]; h->n_q = s.n_q; h->n_lim = s.n_lim;
    h->n_master = s.n_master; h->num_patches = s.num_patches;
    h->bs_limiter_gains = bs_limiter_gains;
    h->bs_interpol_freq = bs_interpol_freq;
    h->bs_smoothing_mode = bs_smoothing_mode;
    h->bs_amp_res_header = bs_amp_res_header;
    for (i = 0; i < 6; i++) {
        h->patch_num_subbands[i] = s.patch_num_subbands[i];
        h->patch_start_subband[i] = s.patch_start_subband[i];
        h->f_tablenoise[i] = s.f_tablenoise[i];
    }
    for (i = 0; i < 25; i++) h->f_tablelow[i] = s.f_tablelow[i];
    for (i = 0; i < 49; i++) h->f_tablehigh[i] = s.f_tablehigh[i];
    for (i = 0; i < 29; i++) h->f_tablelim[i] = s.f_tablelim[i];
    return 0;
}
