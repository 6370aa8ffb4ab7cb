/* validate.h -- record validation at the C-ABI boundary, one implementation for the host
 * (single-frame codec path, unit tests) and the device (heaac_he_check_batch's kernel).
 *
 * The batched entry points take records from any parser (heaac_parse.h is one), so the records are where
 * malformed data would arrive.  The rules are the reference parser's own rejections --
 * read_sbr_grid (aacsbr.c:609-745), sbr_make_f_master / sbr_make_f_derived (:296-593),
 * ff_ps_read_data / read_iid_data / read_icc_data (aacps.c:84-147, 150-279) -- plus the bounds the
 * kernels index with (array sizes of sbr.h / aacps.h).  A record that passes cannot drive an
 * out-of-bounds access; heaac_he_decode_batch additionally clamps the few fields that form
 * global addresses, so an unchecked bad record yields wrong audio, never a fault.
 */
#ifndef HEAAC_VALIDATE_H
#define HEAAC_VALIDATE_H
#include <stddef.h>
#include <stdint.h>
#include "heaac_dsp.h"

#if defined(__HIPCC__)
#define HEAAC_HD __host__ __device__ static inline
#else
#define HEAAC_HD static inline
#endif

HEAAC_HD int heaac_table_ok(const uint8_t *t, int n, int first, int last)
{
    if (t[0] != first || t[n] != last) return 0;
    for (int i = 0; i < n; i++)
        if (t[i] >= t[i + 1]) return 0;
    return 1;
}

/* A header as heaac_sbr_make_header() builds it.  The null header (kx = 32, m = 0: the state
 * before any header, aacsbr.c:130) is valid for frames with start = 0 only. */
HEAAC_HD int heaac_check_sbr_header(const HeaacSbrHeader *h)
{
    const int kx = h->kx, m = h->m, top = kx + m;
    if (kx > 32 || m > 48 || top > 64 || h->k0 > 32 || h->k0 > kx)
        return HEAAC_BAD_HDR_RANGE;
    if (h->bs_limiter_gains > 3 || h->bs_interpol_freq > 1 || h->bs_smoothing_mode > 1 || h->bs_amp_res_header > 1)
        return HEAAC_BAD_HDR_FLAGS;
    if (m == 0)
        return HEAAC_BAD_NONE;              /* nothing below is read without an SBR range */
    if (h->n[0] < 1 || h->n[0] > 24 || h->n[1] < 1 || h->n[1] > 48 || h->n_q < 1 || h->n_q > 5 ||
        h->n_lim < 1 || h->n_lim > 29 || h->num_patches < 1 || h->num_patches > 5)
        return HEAAC_BAD_HDR_COUNTS;
    if (!heaac_table_ok(h->f_tablehigh, h->n[1], kx, top) || !heaac_table_ok(h->f_tablelow, h->n[0], kx, top))
        return HEAAC_BAD_HDR_TABLE;
    /* noise table: a subset of the low-resolution borders; two of them may coincide (:457-461) */
    if (h->f_tablenoise[0] != kx || h->f_tablenoise[h->n_q] != top)
        return HEAAC_BAD_HDR_TABLE;
    for (int i = 0; i < h->n_q; i++)
        if (h->f_tablenoise[i] > h->f_tablenoise[i + 1]) return HEAAC_BAD_HDR_TABLE;
    /* limiter table: from kx over the patched range (the last patch may have been dropped, :538) */
    if (h->f_tablelim[0] != kx || h->f_tablelim[h->n_lim] > top)
        return HEAAC_BAD_HDR_TABLE;
    for (int i = 0; i < h->n_lim; i++)
        if (h->f_tablelim[i] >= h->f_tablelim[i + 1]) return HEAAC_BAD_HDR_TABLE;
    for (int k = kx; k < top; k++) {
        if (h->map_hi[k] >= h->n[1] || h->map_lo[k] >= h->n[0] || h->map_nq[k] >= h->n_q)
            return HEAAC_BAD_HDR_MAP;
        if (h->map_lim[k] != 0xff && h->map_lim[k] >= h->n_lim) return HEAAC_BAD_HDR_MAP;
        if (h->map_mid[k] != 0xff && h->map_mid[k] >= h->n[1]) return HEAAC_BAD_HDR_MAP;
        if (h->map_src[k] != 0xff && h->map_src[k] >= kx) return HEAAC_BAD_HDR_MAP;
    }
    return HEAAC_BAD_NONE;
}

HEAAC_HD int heaac_check_sbr_channel(const HeaacSbrChannel *c, int n_q)
{
    const int L = c->bs_num_env;
    if (L < 1 || L > 5 || c->bs_num_noise != (L > 1) + 1)
        return HEAAC_BAD_SBR_NUM_ENV;
    /* read_sbr_grid: t_env[0] in 0..3, trailing border 16..19, borders increasing
     * (the reference lets two borders coincide, :716; an envelope of no slots divides by zero in
     * sbr_env_estimate, so equality is rejected here) */
    if (c->t_env[0] > 3 || c->t_env[L] < 16 || c->t_env[L] > 19)
        return HEAAC_BAD_SBR_T_ENV;
    for (int i = 0; i < L; i++)
        if (c->t_env[i] >= c->t_env[i + 1]) return HEAAC_BAD_SBR_T_ENV;
    if (c->t_q[0] != c->t_env[0] || c->t_q[c->bs_num_noise] != c->t_env[L])
        return HEAAC_BAD_SBR_T_Q;
    /* The middle noise border is an envelope border of this frame -- or, for a frame with a variable trailing
     * end and bs_pointer = 0, the entry behind the last border that an EARLIER frame left in t_env[] (the
     * reference's unsigned pointer arithmetic, aacsbr.c:729; csrc/sbr_parse.c grid_noise_border_index): any time
     * slot a border can have, or 0.  It only enters `t_env[e] >= t_q[1]` (sbr_mapping, aacsbr.c:1467). */
    if (c->bs_num_noise > 1 && c->t_q[1] > 19) return HEAAC_BAD_SBR_T_Q;
    if (c->bs_amp_res > 1 || c->bs_add_harmonic_flag > 1 || c->t_env_num_env_old > 19 ||
        c->e_a[0] < -1 || c->e_a[0] > 0 || c->e_a[1] < -1 || c->e_a[1] > L)
        return HEAAC_BAD_SBR_FLAGS;
    for (int i = 0; i <= L; i++)
        if (c->bs_freq_res[i] > 1) return HEAAC_BAD_SBR_FLAGS;
    for (int i = 0; i < n_q; i++)
        if (c->bs_invf_mode[0][i] > 3 || c->bs_invf_mode[1][i] > 3) return HEAAC_BAD_SBR_FLAGS;
    return HEAAC_BAD_NONE;
}

/* ncore: SBR channels of the element (2 for a CPE) */
HEAAC_HD int heaac_check_sbr_frame(const HeaacSbrFrame *f, const HeaacSbrHeader *hdr_tab, size_t n_hdr, int ncore)
{
    if (f->hdr >= n_hdr)
        return HEAAC_BAD_HDR_INDEX;
    const HeaacSbrHeader *h = &hdr_tab[f->hdr];
    int r = heaac_check_sbr_header(h);
    if (r) return r;
    if (f->start > 1 || f->reset > 1 || f->bs_coupling > 1 || (f->bs_coupling && ncore != 2))
        return HEAAC_BAD_SBR_FLAGS;
    if (f->kx_old > 32 || f->kx_old + f->m_old > 64)
        return HEAAC_BAD_SBR_OLD_RANGE;
    for (int ch = 0; ch < ncore; ch++)
        if (f->ch[ch].t_env_num_env_old > 19) return HEAAC_BAD_SBR_OLD_RANGE;
    if (!f->start)
        return HEAAC_BAD_NONE;              /* nothing of the channel records is read */
    if (h->m == 0)
        return HEAAC_BAD_HDR_UNSTARTED;
    for (int ch = 0; ch < ncore; ch++) {
        r = heaac_check_sbr_channel(&f->ch[ch], h->n_q);
        if (r) return r;
    }
    return HEAAC_BAD_NONE;
}

HEAAC_HD int heaac_check_ps_frame(const HeaacPsFrame *p)
{
    if (p->start > 1 || p->is34bands > 1 || p->is34bands_old > 1 || p->enable_ipdopd > 1 || p->iid_quant > 1)
        return HEAAC_BAD_PS_NR_PAR;
    if (!p->start)
        return HEAAC_BAD_NONE;              /* mono copy: nothing else is read */
    const int E = p->num_env;
    if (E < 1 || E > 5 || p->num_env_old > 5)
        return HEAAC_BAD_PS_NUM_ENV;
    /* ff_ps_read_data writes -1 first and ends the grid at slot 31 (:186-194, 236-254); explicit borders
     * (frame_class 1) are not checked for order there -- the kernels walk them once, ascending */
    if (p->border_position[0] != -1 || p->border_position[E] != 31)
        return HEAAC_BAD_PS_BORDER;
    for (int e = 0; e < E; e++)
        if (p->border_position[e] >= p->border_position[e + 1]) return HEAAC_BAD_PS_BORDER;
    const int ni = p->nr_iid_par, nc = p->nr_icc_par, np = p->nr_ipdopd_par;
    if ((ni != 10 && ni != 20 && ni != 34) || (nc != 10 && nc != 20 && nc != 34) ||
        (np != 5 && np != 11 && np != 17) || p->icc_mode > 5)
        return HEAAC_BAD_PS_NR_PAR;
    const int lim = 7 + 8 * p->iid_quant;
    for (int e = 0; e < E; e++) {
        for (int b = 0; b < ni; b++)
            if (p->iid_par[e][b] < -lim || p->iid_par[e][b] > lim) return HEAAC_BAD_PS_PAR;
        for (int b = 0; b < nc; b++)
            if (p->icc_par[e][b] < 0 || p->icc_par[e][b] > 7) return HEAAC_BAD_PS_PAR;
        if (p->enable_ipdopd)
            for (int b = 0; b < np; b++)
                if (p->ipd_par[e][b] < 0 || p->ipd_par[e][b] > 7 || p->opd_par[e][b] < 0 || p->opd_par[e][b] > 7)
                    return HEAAC_BAD_PS_PAR;
    }
    return HEAAC_BAD_NONE;
}

#endif
