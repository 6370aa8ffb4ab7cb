// k_ps.hip -- Parametric Stereo for gfx950: ff_ps_apply() (aacps.c:973-992) =
// hybrid_analysis (:359-395), decorrelation (:645-754), stereo_processing
// (:794-971), hybrid_synthesis (:397-445), one wavefront per frame.
//
// Lane = frequency band.  Every recursion of the reference (transient smoother,
// all-pass chain, H-matrix interpolation) runs over time inside one lane, so
// the reference's operation order is kept and parallelism comes from the
// 71/91 hybrid bands.  Two passes: pass A handles the bands that are plain QMF
// bands (read and written as coalesced rows of X), pass B the hybrid
// sub-subbands of the lowest 3/5 QMF bands (kept in LDS).
#include "k_common.h"
#include "kernels.h"

#define PS_WAVES 4
#define SUB_STRIDE 66          // one sub-subband row: 32 slots * (re,im) + 2 pad
#define PN_STRIDE 92

__device__ const signed char k_to_i_20_d[71] = {
     1,  0,  0,  1,  2,  3,  4,  5,  6,  7,  8,  9, 10, 11, 12, 13, 14, 14, 15,
    15, 15, 16, 16, 16, 16, 17, 17, 17, 17, 17, 18, 18, 18, 18, 18, 18, 18, 18,
    18, 18, 18, 18, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19,
    19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19
};
__device__ const signed char k_to_i_34_d[91] = {
     0,  1,  2,  3,  4,  5,  6,  6,  7,  2,  1,  0, 10, 10,  4,  5,  6,  7,  8,
     9, 10, 11, 12,  9, 14, 11, 12, 13, 14, 15, 16, 13, 16, 17, 18, 19, 20, 21,
    22, 22, 23, 23, 24, 24, 25, 25, 26, 26, 27, 27, 27, 28, 28, 28, 29, 29, 29,
    30, 30, 30, 31, 31, 31, 31, 32, 32, 32, 32, 33, 33, 33, 33, 33, 33, 33, 33,
    33, 33, 33, 33, 33, 33, 33, 33, 33, 33, 33, 33, 33, 33, 33
};

struct PsWave {
    HeaacPsFrame p;
    float inb[5][44][2];               // hybrid analysis input: 6 history + 38 current slots
    float sub[32][SUB_STRIDE];         // sub-subband signals s[ks][n]; L output in place
    float pn[32 * PN_STRIDE];          // |s|^2 [n][k]; later R output of the sub-subbands
    float pw[34][33];                  // band power, then transient gain
    float Hs[6][8][34];                // H11r,H11i,H12r,H12i,H21r,H21i,H22r,H22i rows per envelope border
    signed char kti[92];               // k_to_i for this frame's band layout
    signed char iid_m[5][34], icc_m[5][34], ipd_m[5][34], opd_m[5][34];
};

// map_idx_* (aacps.c:461-643) as a gather: mapped value of band b.
__device__ __forceinline__ int remap_idx(const signed char *par, int num_par, int to34, int b)
{
    if (to34) {
        if (num_par == 20 || num_par == 11) {
            // map_idx_20_to_34
            const signed char src[34] = { 0, -1, 1, 2, -2, 3, 4, 4, 5, 5, 6, 7, 8, 8, 9, 9, 10,
                                          11, 12, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18, 18, 18, 19, 19 };
            const int s = src[b];
            if (s == -1) return (par[0] + par[1]) / 2;
            if (s == -2) return (par[2] + par[3]) / 2;
            return par[s];
        }
        if (num_par == 10 || num_par == 5) {
            const signed char src[34] = { 0, 0, 0, 1, 1, 1, 2, 2, 2, 2, 3, 3, 4, 4, 4, 4, 5, 5, 6, 6,
                                          7, 7, 7, 7, 8, 8, 8, 8, 9, 9, 9, 9, 9, 9 };
            if (num_par == 5 && b >= 16) return 0;            // full == 0: par_mapped[16] = 0
            return par[src[b]];
        }
        return par[b];
    }
    if (num_par == 34 || num_par == 17) {
        // map_idx_34_to_20
        switch (b) {
        case 0:  return (2 * par[0] + par[1]) / 3;
        case 1:  return (par[1] + 2 * par[2]) / 3;
        case 2:  return (2 * par[3] + par[4]) / 3;
        case 3:  return (par[4] + 2 * par[5]) / 3;
        case 4:  return (par[6] + par[7]) / 2;
        case 5:  return (par[8] + par[9]) / 2;
        case 6:  return par[10];
        case 7:  return par[11];
        case 8:  return (par[12] + par[13]) / 2;
        case 9:  return (par[14] + par[15]) / 2;
        case 10: return par[16];
        case 11: return par[17];
        case 12: return par[18];
        case 13: return par[19];
        case 14: return (par[20] + par[21]) / 2;
        case 15: return (par[22] + par[23]) / 2;
        case 16: return (par[24] + par[25]) / 2;
        case 17: return (par[26] + par[27]) / 2;
        case 18: return (par[28] + par[29] + par[30] + par[31]) / 4;
        case 19: return (par[32] + par[33]) / 2;
        }
        return 0;
    }
    if (num_par == 10 || num_par == 5) {
        if (num_par == 5 && b >= 10) return 0;                // full == 0: par_mapped[10] = 0
        return par[b >> 1];
    }
    return par[b];
}

// map_val_20_to_34 / map_val_34_to_20 (aacps.c:491-514, 598-634) as gathers.
__device__ __forceinline__ float remap_val(const float *par, int to34, int b)
{
    if (to34) {
        const signed char src[34] = { 0, -1, 1, 2, -2, 3, 4, 4, 5, 5, 6, 7, 8, 8, 9, 9, 10,
                                      11, 12, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18, 18, 18, 19, 19 };
        const int s = src[b];
        if (s == -1) return (par[0] + par[1]) * 0.5f;
        if (s == -2) return (par[2] + par[3]) * 0.5f;
        return par[s];
    }
    switch (b) {
    case 0:  return (2 * par[0] + par[1]) * 0.33333333f;
    case 1:  return (par[1] + 2 * par[2]) * 0.33333333f;
    case 2:  return (2 * par[3] + par[4]) * 0.33333333f;
    case 3:  return (par[4] + 2 * par[5]) * 0.33333333f;
    case 4:  return (par[6] + par[7]) * 0.5f;
    case 5:  return (par[8] + par[9]) * 0.5f;
    case 6:  return par[10];
    case 7:  return par[11];
    case 8:  return (par[12] + par[13]) * 0.5f;
    case 9:  return (par[14] + par[15]) * 0.5f;
    case 10: return par[16];
    case 11: return par[17];
    case 12: return par[18];
    case 13: return par[19];
    case 14: return (par[20] + par[21]) * 0.5f;
    case 15: return (par[22] + par[23]) * 0.5f;
    case 16: return (par[24] + par[25]) * 0.5f;
    case 17: return (par[26] + par[27]) * 0.5f;
    case 18: return (par[28] + par[29] + par[30] + par[31]) * 0.25f;
    case 19: return (par[32] + par[33]) * 0.5f;
    }
    return par[b];          // 34 -> 20 leaves par[20..33] untouched
}

// 13-tap complex FIR of hybrid6_cx / hybrid4_8_12_cx (aacps.c:310-321, :343-353).
// in: 13 consecutive complex slots, filt: [7][2]
__device__ __forceinline__ void hybrid_fir(const float *in, const float *filt, float &o_re, float &o_im)
{
    float sum_re = filt[12] * in[12], sum_im = filt[12] * in[13];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const float in0_re = in[2 * j], in0_im = in[2 * j + 1];
        const float in1_re = in[2 * (12 - j)], in1_im = in[2 * (12 - j) + 1];
        sum_re += filt[2 * j] * (in0_re + in1_re) - filt[2 * j + 1] * (in0_im - in1_im);
        sum_im += filt[2 * j] * (in0_im + in1_im) + filt[2 * j + 1] * (in0_re - in1_re);
    }
    o_re = sum_re;
    o_im = sum_im;
}

// One band, all 32 slots: decorrelation (aacps.c:696-753) fused with the mixing
// loop of stereo_processing (:900-969).
//   src(n, re, im)        : s[k][n]
//   sink(n, lre, lim, rre, rim)
//   dl / ap               : state pointers already offset to this band's column
template <class Src, class Sink>
__device__ __forceinline__ void ps_band(PsWave &w, const float *__restrict__ g_tab, int is34, int kh,
                                        bool clear_delay, bool clear_ap,
                                        const float *dl_in, float *dl_out, int dl_stride,
                                        const float *ap_in, float *ap_out, int ap_stride,
                                        Src src, Sink sink)
{
    const int nr_allpass = is34 ? 50 : 30, short_delay = is34 ? 62 : 42;
    const int b = w.kti[kh];
    const int enable_ipdopd = w.p.enable_ipdopd;

    // delay line tail d[j] = s[k][j - 14], j = 0..13
    float dre[14], dim[14];
#pragma unroll
    for (int j = 0; j < 14; j++) {
        dre[j] = clear_delay ? 0.0f : dl_in[j * dl_stride];
        dim[j] = clear_delay ? 0.0f : dl_in[j * dl_stride + 1];
    }
    const bool allpass = kh < nr_allpass;
    const bool d14 = !allpass && kh < short_delay;

    // all-pass state: last 5 values of every link
    float are[3][5], aim[3][5];
    float ag[3], qre[3], qim[3], phre = 0.0f, phim = 0.0f;
    if (allpass) {
        float g_decay_slope = 1.f - 0.05f * (kh - (is34 ? 32 : 10));
        g_decay_slope = g_decay_slope < 0.f ? 0.f : (g_decay_slope > 1.f ? 1.f : g_decay_slope);   // av_clipf
        const float a[3] = { 0.65143905753106f, 0.56471812200776f, 0.48954165955695f };
#pragma unroll
        for (int m = 0; m < 3; m++) {
            ag[m] = a[m] * g_decay_slope;
            qre[m] = g_tab[TB_QFRACT + ((is34 * 50 + kh) * 3 + m) * 2];
            qim[m] = g_tab[TB_QFRACT + ((is34 * 50 + kh) * 3 + m) * 2 + 1];
#pragma unroll
            for (int j = 0; j < 5; j++) {
                are[m][j] = clear_ap ? 0.0f : ap_in[(m * 5 + j) * ap_stride];
                aim[m][j] = clear_ap ? 0.0f : ap_in[(m * 5 + j) * ap_stride + 1];
            }
        }
        phre = g_tab[TB_PHIFRACT + (is34 * 50 + kh) * 2];
        phim = g_tab[TB_PHIFRACT + (is34 * 50 + kh) * 2 + 1];
    }

    const bool neg_im = (is34 && kh <= 13 && kh >= 9) || (!is34 && kh <= 1);

    for (int e = 0; e < w.p.num_env; e++) {
        const int start = w.p.border_position[e], stop = w.p.border_position[e + 1];
        const float width = 1.f / (stop - start);
        float h11r = w.Hs[e][0][b], h12r = w.Hs[e][2][b], h21r = w.Hs[e][4][b], h22r = w.Hs[e][6][b];
        float h11i = 0, h12i = 0, h21i = 0, h22i = 0;
        float h11i_step = 0, h12i_step = 0, h21i_step = 0, h22i_step = 0;
        if (enable_ipdopd) {
            h11i = w.Hs[e][1][b]; h12i = w.Hs[e][3][b]; h21i = w.Hs[e][5][b]; h22i = w.Hs[e][7][b];
            if (neg_im) { h11i = -h11i; h12i = -h12i; h21i = -h21i; h22i = -h22i; }
        }
        const float h11r_step = (w.Hs[e + 1][0][b] - h11r) * width;
        const float h12r_step = (w.Hs[e + 1][2][b] - h12r) * width;
        const float h21r_step = (w.Hs[e + 1][4][b] - h21r) * width;
        const float h22r_step = (w.Hs[e + 1][6][b] - h22r) * width;
        if (enable_ipdopd) {
            h11i_step = (w.Hs[e + 1][1][b] - h11i) * width;
            h12i_step = (w.Hs[e + 1][3][b] - h12i) * width;
            h21i_step = (w.Hs[e + 1][5][b] - h21i) * width;
            h22i_step = (w.Hs[e + 1][7][b] - h22i) * width;
        }
        for (int n = start + 1; n <= stop; n++) {
            float sre, sim;
            src(n, sre, sim);
            const float tg = w.pw[b][n];
            float r_re, r_im;
            if (allpass) {
                // z^-2 tap (delay[k][n + PS_MAX_DELAY - 2]) times phi_fract
                float in_re = dre[12] * phre - dim[12] * phim;
                float in_im = dre[12] * phim + dim[12] * phre;
#pragma unroll
                for (int m = 0; m < 3; m++) {
                    const float a_re = ag[m] * in_re, a_im = ag[m] * in_im;
                    // link_delay = 3, 4, 5 -> history position 5 - delay
                    const float ld_re = are[m][2 - m], ld_im = aim[m][2 - m];
                    float nre = in_re, nim = in_im;
                    in_re = ld_re * qre[m] - ld_im * qim[m] - a_re;
                    in_im = ld_re * qim[m] + ld_im * qre[m] - a_im;
                    nre += ag[m] * in_re;
                    nim += ag[m] * in_im;
#pragma unroll
                    for (int j = 0; j < 4; j++) { are[m][j] = are[m][j + 1]; aim[m][j] = aim[m][j + 1]; }
                    are[m][4] = nre; aim[m][4] = nim;
                }
                r_re = tg * in_re;
                r_im = tg * in_im;
            } else if (d14) {
                r_re = tg * dre[0];          // delay[k][n + PS_MAX_DELAY - 14]
                r_im = tg * dim[0];
            } else {
                r_re = tg * dre[13];         // delay[k][n + PS_MAX_DELAY - 1]
                r_im = tg * dim[13];
            }
            // advance the 14-slot delay line (the reference keeps 46 slots and
            // reads at an offset; a register shift is the same data flow)
#pragma unroll
            for (int j = 0; j < 13; j++) { dre[j] = dre[j + 1]; dim[j] = dim[j + 1]; }
            dre[13] = sre; dim[13] = sim;

            h11r += h11r_step; h12r += h12r_step; h21r += h21r_step; h22r += h22r_step;
            float lre, lim, rre, rim;
            if (enable_ipdopd) {
                h11i += h11i_step; h12i += h12i_step; h21i += h21i_step; h22i += h22i_step;
                lre = h11r * sre + h21r * r_re - h11i * sim - h21i * r_im;
                lim = h11r * sim + h21r * r_im + h11i * sre + h21i * r_re;
                rre = h12r * sre + h22r * r_re - h12i * sim - h22i * r_im;
                rim = h12r * sim + h22r * r_im + h12i * sre + h22i * r_re;
            } else {
                lre = h11r * sre + h21r * r_re;
                lim = h11r * sim + h21r * r_im;
                rre = h12r * sre + h22r * r_re;
                rim = h12r * sim + h22r * r_im;
            }
            sink(n, lre, lim, rre, rim);
        }
    }
#pragma unroll
    for (int j = 0; j < 14; j++) {
        dl_out[j * dl_stride]     = dre[j];
        dl_out[j * dl_stride + 1] = dim[j];
    }
    if (allpass) {
#pragma unroll
        for (int m = 0; m < 3; m++)
#pragma unroll
            for (int j = 0; j < 5; j++) {
                ap_out[(m * 5 + j) * ap_stride]     = are[m][j];
                ap_out[(m * 5 + j) * ap_stride + 1] = aim[m][j];
            }
    }
}

__device__ __forceinline__ void ps_frame(PsWave &w, const float *__restrict__ g_tab,
                                         const HeaacPsFrame *g_p, int top_qmf,
                                         const float *st_in, float *st_out,
                                         float *XL /* [2][38][64] in: mono, out: left */,
                                         float *XR /* [2][38][64] out: right */, int lane)
{
    {
        const uint32_t *s = reinterpret_cast<const uint32_t *>(g_p);
        uint32_t *d = reinterpret_cast<uint32_t *>(&w.p);
        for (int i = lane; i < (int)(sizeof(HeaacPsFrame) / 4); i += WAVE) d[i] = s[i];
    }
    wave_sync();
    const HeaacPsFrame &p = w.p;
    float *XL0 = XL, *XL1 = XL + 38 * 64, *XR0 = XR, *XR1 = XR + 38 * 64;

    if (!p.start) {
        // memcpy(sbr->X[1], sbr->X[0]) (aacsbr.c:1755); PS state untouched
        for (int t = lane; t < 2 * 38 * 64; t += WAVE) XR[t] = XL[t];
        if (st_out != st_in)
            for (int t = lane; t < HEAAC_ST_PS; t += WAVE) st_out[t] = st_in[t];
        wave_sync();
        return;
    }

    const int is34 = p.is34bands;
    const int nr_bands = is34 ? 91 : 71, nr_par = is34 ? 34 : 20, nr_allpass = is34 ? 50 : 30;
    const int nsub = is34 ? 32 : 10, nlow = is34 ? 5 : 3;     // sub-subbands / hybrid QMF bands
    const int top = top_qmf + nr_bands - 64;                  // aacps.c:980
    const bool switched = is34 != p.is34bands_old;

    for (int k = lane; k < nr_bands; k += WAVE) w.kti[k] = is34 ? k_to_i_34_d[k] : k_to_i_20_d[k];

    // ---- hybrid analysis input (aacps.c:362-367): in[i][j+6] = L[.][j][i] ----
    for (int t = lane; t < 5 * 6; t += WAVE) {
        const int i = t / 6, j = t % 6;
        w.inb[i][j][0] = st_in[HEAAC_PS_INBUF + t * 2];
        w.inb[i][j][1] = st_in[HEAAC_PS_INBUF + t * 2 + 1];
    }
    for (int t = lane; t < 5 * 38; t += WAVE) {
        const int i = t % 5, j = t / 5;
        w.inb[i][j + 6][0] = XL0[j * 64 + i];
        w.inb[i][j + 6][1] = XL1[j * 64 + i];
    }
    wave_sync();
    // in_buf update (:391-394): in[i][0..5] <- in[i][32..37]
    for (int t = lane; t < 5 * 6; t += WAVE) {
        const int i = t / 6, j = t % 6;
        st_out[HEAAC_PS_INBUF + t * 2]     = w.inb[i][32 + j][0];
        st_out[HEAAC_PS_INBUF + t * 2 + 1] = w.inb[i][32 + j][1];
    }
    // ---- hybrid filters -> sub[ks][n] ----
    for (int t = lane; t < nsub * 32; t += WAVE) {
        const int ks = t >> 5, n = t & 31;
        float re, im;
        if (is34) {
            int qb, f, off;
            if (ks < 12)      { qb = 0; f = ks;      off = TB_F34_0_12; }
            else if (ks < 20) { qb = 1; f = ks - 12; off = TB_F34_1_8; }
            else              { qb = 2 + ((ks - 20) >> 2); f = (ks - 20) & 3; off = TB_F34_2_4; }
            hybrid_fir(&w.inb[qb][n][0], g_tab + off + f * 14, re, im);
        } else if (ks < 6) {
            // hybrid6_cx output order (:322-334)
            const float *in = &w.inb[0][n][0];
            const float *F = g_tab + TB_F20_0_8;
            if (ks == 0)      hybrid_fir(in, F + 6 * 14, re, im);
            else if (ks == 1) hybrid_fir(in, F + 7 * 14, re, im);
            else if (ks == 2) hybrid_fir(in, F + 0 * 14, re, im);
            else if (ks == 3) hybrid_fir(in, F + 1 * 14, re, im);
            else {
                float ar, ai, br, bi;
                hybrid_fir(in, F + (ks == 4 ? 2 : 3) * 14, ar, ai);
                hybrid_fir(in, F + (ks == 4 ? 5 : 4) * 14, br, bi);
                re = ar + br;
                im = ai + bi;
            }
        } else {
            // hybrid2_re (:283-301): band 1 reversed, band 2 not
            const int qb = ks < 8 ? 1 : 2, reverse = ks < 8 ? 1 : 0;
            const int which = (ks - (qb == 1 ? 6 : 8));       // 0 -> out[0], 1 -> out[1]
            const float *in = &w.inb[qb][n][0];
            const float *f = g_tab + TB_G1_Q2;
            const float re_in = f[6] * in[12], im_in = f[6] * in[13];
            float re_op = 0.0f, im_op = 0.0f;
#pragma unroll
            for (int j = 0; j < 6; j += 2) {
                re_op += f[j + 1] * (in[2 * (j + 1)] + in[2 * (12 - j - 1)]);
                im_op += f[j + 1] * (in[2 * (j + 1) + 1] + in[2 * (12 - j - 1) + 1]);
            }
            // out[reverse] = in + op, out[!reverse] = in - op
            if (which == reverse) { re = re_in + re_op; im = im_in + im_op; }
            else                  { re = re_in - re_op; im = im_in - im_op; }
        }
        w.sub[ks][2 * n] = re;
        w.sub[ks][2 * n + 1] = im;
        w.pn[n * PN_STRIDE + ks] = re * re + im * im;
    }
    // |s|^2 of the plain QMF bands
    for (int n = 0; n < 32; n++) {
        const int q = lane;
        if (q >= nlow) {
            const float re = XL0[n * 64 + q], im = XL1[n * 64 + q];
            w.pn[n * PN_STRIDE + q - nlow + nsub] = re * re + im * im;
        }
    }
    wave_sync();

    // ---- band power (aacps.c:673-678): ascending k per parameter band ----
    for (int t = lane; t < nr_par * 32; t += WAVE) {
        const int i = t >> 5, n = t & 31;
        float acc = 0.0f;
        for (int k = 0; k < nr_bands; k++)
            if (w.kti[k] == i)
                acc += w.pn[n * PN_STRIDE + k];
        w.pw[i][n] = acc;
    }
    wave_sync();
    // ---- transient detection (:681-692), one lane per parameter band ----
    if (lane < nr_par) {
        const int i = lane;
        float peak = switched ? 0.0f : st_in[HEAAC_PS_PEAK + i];
        float smooth = switched ? 0.0f : st_in[HEAAC_PS_PSMOOTH + i];
        float diff = switched ? 0.0f : st_in[HEAAC_PS_PDIFF + i];
        for (int n = 0; n < 32; n++) {
            const float pwr = w.pw[i][n];
            const float decayed_peak = 0.76592833836465f * peak;
            peak = decayed_peak > pwr ? decayed_peak : pwr;
            smooth += 0.25f * (pwr - smooth);
            diff += 0.25f * (peak - pwr - diff);
            const float denom = 1.5f * diff;
            w.pw[i][n] = (denom > smooth) ? smooth / denom : 1.0f;
        }
        st_out[HEAAC_PS_PEAK + i] = peak;
        st_out[HEAAC_PS_PSMOOTH + i] = smooth;
        st_out[HEAAC_PS_PDIFF + i] = diff;
    } else if (lane < 34) {
        // parameter bands 20..33 are not touched in 20-band mode
        const int i = lane;
        st_out[HEAAC_PS_PEAK + i]    = switched ? 0.0f : st_in[HEAAC_PS_PEAK + i];
        st_out[HEAAC_PS_PSMOOTH + i] = switched ? 0.0f : st_in[HEAAC_PS_PSMOOTH + i];
        st_out[HEAAC_PS_PDIFF + i]   = switched ? 0.0f : st_in[HEAAC_PS_PDIFF + i];
    }

    // ---- parameter remapping + H matrices (aacps.c:817-899) ----
    for (int t = lane; t < 5 * 34; t += WAVE) {
        const int e = t / 34, b = t % 34;
        int iid = 0, icc = 0, ipd = 0, opd = 0;
        if (e < p.num_env && b < nr_par) {
            iid = remap_idx(p.iid_par[e], p.nr_iid_par, is34, b);
            icc = remap_idx(p.icc_par[e], p.nr_icc_par, is34, b);
            if (p.enable_ipdopd && b < 17) {
                ipd = remap_idx(p.ipd_par[e], p.nr_ipdopd_par, is34, b);
                opd = remap_idx(p.opd_par[e], p.nr_ipdopd_par, is34, b);
            }
        }
        w.iid_m[e][b] = (signed char)iid; w.icc_m[e][b] = (signed char)icc;
        w.ipd_m[e][b] = (signed char)ipd; w.opd_m[e][b] = (signed char)opd;
    }
    // row 0 = H of the last envelope of the previous frame, remapped on a 20<->34 switch
    for (int t = lane; t < 8 * 34; t += WAVE) {
        const int j = t / 34, b = t % 34;
        const float *row = st_in + HEAAC_PS_H + j * 34;
        w.Hs[0][j][b] = switched ? remap_val(row, is34, b) : row[b];
    }
    wave_sync();
    if (lane < nr_par) {
        const int b = lane;
        const float *LUT = g_tab + ((p.icc_mode < 3) ? TB_HA : TB_HB);
        const signed char *hist = reinterpret_cast<const signed char *>(st_in + HEAAC_PS_HIST);
        int opd_hist = hist[b], ipd_hist = hist[34 + b];
        if (switched && b < 17) { opd_hist = 0; ipd_hist = 0; }        // ipdopd_reset
        for (int e = 0; e < p.num_env; e++) {
            const float *h = LUT + ((w.iid_m[e][b] + 7 + 23 * p.iid_quant) * 8 + w.icc_m[e][b]) * 4;
            float h11 = h[0], h12 = h[1], h21 = h[2], h22 = h[3];
            float h11i = 0.0f, h12i = 0.0f, h21i = 0.0f, h22i = 0.0f;
            if (p.enable_ipdopd && b < p.nr_ipdopd_par) {
                const int opd_idx = opd_hist * 8 + w.opd_m[e][b];
                const int ipd_idx = ipd_hist * 8 + w.ipd_m[e][b];
                const float opd_re = g_tab[TB_PD_RE + opd_idx], opd_im = g_tab[TB_PD_IM + opd_idx];
                const float ipd_re = g_tab[TB_PD_RE + ipd_idx], ipd_im = g_tab[TB_PD_IM + ipd_idx];
                opd_hist = opd_idx & 0x3F;
                ipd_hist = ipd_idx & 0x3F;
                const float ipd_adj_re = opd_re * ipd_re + opd_im * ipd_im;
                const float ipd_adj_im = opd_im * ipd_re - opd_re * ipd_im;
                h11i = h11 * opd_im;     h11 = h11 * opd_re;
                h12i = h12 * ipd_adj_im; h12 = h12 * ipd_adj_re;
                h21i = h21 * opd_im;     h21 = h21 * opd_re;
                h22i = h22 * ipd_adj_im; h22 = h22 * ipd_adj_re;
            }
            w.Hs[e + 1][0][b] = h11; w.Hs[e + 1][1][b] = h11i;
            w.Hs[e + 1][2][b] = h12; w.Hs[e + 1][3][b] = h12i;
            w.Hs[e + 1][4][b] = h21; w.Hs[e + 1][5][b] = h21i;
            w.Hs[e + 1][6][b] = h22; w.Hs[e + 1][7][b] = h22i;
        }
        // new history (bytes of two packed rows)
        signed char *ho = reinterpret_cast<signed char *>(st_out + HEAAC_PS_HIST);
        ho[b] = (signed char)opd_hist;
        ho[34 + b] = (signed char)ipd_hist;
    } else if (lane < 34) {
        const signed char *hist = reinterpret_cast<const signed char *>(st_in + HEAAC_PS_HIST);
        signed char *ho = reinterpret_cast<signed char *>(st_out + HEAAC_PS_HIST);
        ho[lane] = hist[lane];
        ho[34 + lane] = hist[34 + lane];
    }
    if (lane == 0) {
        signed char *ho = reinterpret_cast<signed char *>(st_out + HEAAC_PS_HIST);
        ho[68] = ho[69] = ho[70] = ho[71] = 0;
    }
    wave_sync();
    // H state out: real rows always, imaginary rows only while IPD/OPD is on
    for (int t = lane; t < 8 * 34; t += WAVE) {
        const int j = t / 34, b = t % 34;
        const bool imag = j & 1;
        float v;
        if (!imag) v = b < nr_par ? w.Hs[p.num_env][j][b] : 0.0f;
        else if (p.enable_ipdopd) v = b < nr_par ? w.Hs[p.num_env][j][b] : 0.0f;
        else v = st_in[HEAAC_PS_H + t];
        st_out[HEAAC_PS_H + t] = v;
    }

    const float *dl_in = st_in + HEAAC_PS_DELAY;
    float *dl_out = st_out + HEAAC_PS_DELAY;
    const float *ap_in = st_in + HEAAC_PS_APDELAY;
    float *ap_out = st_out + HEAAC_PS_APDELAY;

    // ---- pass A: plain QMF bands q >= nlow, hybrid index kh = q - nlow + nsub ----
    if (lane >= nlow) {
        const int q = lane, kh = q - nlow + nsub;
        ps_band(w, g_tab, is34, kh, switched || kh >= top, switched || kh >= top,
                dl_in + kh * 2, dl_out + kh * 2, 91 * 2, ap_in + kh * 2, ap_out + kh * 2, 50 * 2,
                [&](int n, float &re, float &im) { re = XL0[n * 64 + q]; im = XL1[n * 64 + q]; },
                [&](int n, float lre, float lim, float rre, float rim) {
                    XL0[n * 64 + q] = lre; XL1[n * 64 + q] = lim;
                    XR0[n * 64 + q] = rre; XR1[n * 64 + q] = rim;
                });
    }
    // ---- pass B: hybrid sub-subbands ----
    float *subR = w.pn;                  // |s|^2 is dead: reuse as R rows [ks][SUB_STRIDE]
    wave_sync();
    if (lane < nsub) {
        const int kh = lane;
        float *srow = w.sub[kh];
        float *rrow = subR + kh * SUB_STRIDE;
        ps_band(w, g_tab, is34, kh, switched || kh >= top, switched || kh >= top,
                dl_in + kh * 2, dl_out + kh * 2, 91 * 2, ap_in + kh * 2, ap_out + kh * 2, 50 * 2,
                [&](int n, float &re, float &im) { re = srow[2 * n]; im = srow[2 * n + 1]; },
                [&](int n, float lre, float lim, float rre, float rim) {
                    srow[2 * n] = lre; srow[2 * n + 1] = lim;
                    rrow[2 * n] = rre; rrow[2 * n + 1] = rim;
                });
    }
    // bands that exist in the state record but not in this layout / all-pass set
    for (int t = lane; t < 14 * 91; t += WAVE) {
        const int k = t % 91;
        if (k >= nr_bands) {
            dl_out[t * 2]     = switched ? 0.0f : dl_in[t * 2];
            dl_out[t * 2 + 1] = switched ? 0.0f : dl_in[t * 2 + 1];
        }
    }
    for (int t = lane; t < 15 * 50; t += WAVE) {
        const int k = t % 50;
        if (k >= nr_allpass) {
            ap_out[t * 2]     = switched ? 0.0f : ap_in[t * 2];
            ap_out[t * 2 + 1] = switched ? 0.0f : ap_in[t * 2 + 1];
        }
    }
    wave_sync();

    // ---- hybrid synthesis (aacps.c:397-445) for the lowest QMF bands ----
    for (int t = lane; t < 2 * 32; t += WAVE) {
        const int n = t & 31, side = t >> 5;
        const float *rows = side ? subR : &w.sub[0][0];
        float *O0 = side ? XR0 : XL0, *O1 = side ? XR1 : XL1;
#define SUBV(i, c) rows[(i) * SUB_STRIDE + 2 * n + (c)]
        if (is34) {
            const int first[5] = { 0, 12, 20, 24, 28 }, cnt[5] = { 12, 8, 4, 4, 4 };
#pragma unroll
            for (int q = 0; q < 5; q++) {
                float re = 0.0f, im = 0.0f;
                for (int i = 0; i < cnt[q]; i++) { re += SUBV(first[q] + i, 0); im += SUBV(first[q] + i, 1); }
                O0[n * 64 + q] = re;
                O1[n * 64 + q] = im;
            }
        } else {
            O0[n * 64 + 0] = SUBV(0, 0) + SUBV(1, 0) + SUBV(2, 0) + SUBV(3, 0) + SUBV(4, 0) + SUBV(5, 0);
            O1[n * 64 + 0] = SUBV(0, 1) + SUBV(1, 1) + SUBV(2, 1) + SUBV(3, 1) + SUBV(4, 1) + SUBV(5, 1);
            O0[n * 64 + 1] = SUBV(6, 0) + SUBV(7, 0);
            O1[n * 64 + 1] = SUBV(6, 1) + SUBV(7, 1);
            O0[n * 64 + 2] = SUBV(8, 0) + SUBV(9, 0);
            O1[n * 64 + 2] = SUBV(8, 1) + SUBV(9, 1);
        }
#undef SUBV
    }
    wave_sync();
}

__global__ __launch_bounds__(PS_WAVES * WAVE)
void k_ps(const float *__restrict__ g_tab, const HeaacPsFrame *__restrict__ g_ps,
          const HeaacSbrFrame *__restrict__ g_sbr, const HeaacSbrHeader *__restrict__ g_hdr,
          const float *g_state_in, float *g_state_out, int state_words, int off_ps,
          float *g_X, unsigned long long n)
{
    __shared__ PsWave S[PS_WAVES];
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
    for (unsigned long long f = (unsigned long long)blockIdx.x * PS_WAVES + wave; f < n;
         f += (unsigned long long)gridDim.x * PS_WAVES) {
        const HeaacSbrHeader &h = g_hdr[g_sbr[f].hdr];
        const int top = h.kx + h.m;                 // ff_ps_apply(..., sbr->kx[1] + sbr->m[1])
        float *XL = g_X + (f * 2) * (2 * 38 * 64);
        ps_frame(S[wave], g_tab, &g_ps[f], top, g_state_in + f * state_words + off_ps,
                 g_state_out + f * state_words + off_ps, XL, XL + 2 * 38 * 64, lane);
    }
}

extern "C" int heaac_launch_ps(const float *d_tab, const HeaacPsFrame *d_ps, const HeaacSbrFrame *d_sbr,
                               const HeaacSbrHeader *d_hdr, const float *d_state_in, float *d_state_out,
                               int state_words, int off_ps, float *d_ws_X, size_t n, hipStream_t s)
{
    if (!n) return HEAAC_OK;
    unsigned long long g = (n + PS_WAVES - 1) / PS_WAVES;
    if (g > 256) g = 256;
    hipLaunchKernelGGL(k_ps, dim3((unsigned)g), dim3(PS_WAVES * WAVE), 0, s, d_tab, d_ps, d_sbr, d_hdr,
                       d_state_in, d_state_out, state_words, off_ps, d_ws_X, (unsigned long long)n);
    return hipGetLastError() == hipSuccess ? HEAAC_OK : HEAAC_ERR_HIP;
}
