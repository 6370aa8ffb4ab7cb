// k_hfr.h -- the HF stage (sbr_lf_gen .. sbr_x_gen, aacsbr.c:1337-1714) with X_low in REGISTERS: the form the
// twelve-wave fused kernel k_hfps12 (k_ps.hip) runs.  Same arithmetic, statement for statement, as hf_channel
// (k_hf.h); what differs is where the low-band signal lives.
//
// hf_channel keeps X_low[32][40] in 10 KB of LDS per wave -- which, with two waves per SIMD already holding every
// register, is what stops a third wave.  Here lane k holds ITS OWN band's row in 40 register pairs:
//   xl[j] = X_low[k][j]            for the lanes below kx (zero above, as sbr_lf_gen zeroes them)
// and the same registers are the stage's OUTPUT: sbr_x_gen's X[.][i][k] is X_low[k][i + 2] for the low bands --
// already there -- and Y[.][i][k] for the SBR range, which those lanes' own registers are free to take
// (xl[i + 2] = X[.][i][k] for all 38 slots when the stage returns: the PS stage's QMF column).
//   * the inverse filter's autocorrelation (aacsbr.c:1232-1255) runs in the band's own lane over its registers;
//   * a lane of the SBR range reads its patch source band from the SOURCE LANE's registers (ds_bpermute: the LDS
//     crossbar, no LDS memory): once, when sbr_hf_gen's X_high is formed -- it is kept in place in xl[] and both the
//     envelope estimate and sbr_hf_assemble read it from there (hf_channel forms it twice);
//   * W is loaded column-wise (lane = band), the previous frame's tail in front of it.
// Only what the headline path needs is here: one channel, no coupling, interpolated envelope estimate
// (bs_interpol_freq = 1).  Frames outside that are left to k_hfps (hf_frame_is_fast).
#pragma once
#include "k_hf.h"

#define HFR_REC_WORDS ((int)((sizeof(HeaacSbrHeader) + sizeof(HeaacSbrChannel)) / 4))

// value of `v` in the lane whose byte address (lane * 4) is `addr`
__device__ __forceinline__ float lane_get(int addr, float v)
{
    return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v)));
}
__device__ __forceinline__ v2f lane_get2(int addr, v2f v)
{
    return v2f{lane_get(addr, v.x), lane_get(addr, v.y)};
}

// Which frames the register form takes (wave-uniform; g_fr / the header are read through the scalar cache).
__device__ __forceinline__ bool hf_frame_is_fast(const HeaacSbrFrame *g_fr, const HeaacSbrHeader *g_hdr, unsigned n_hdr)
{
    const unsigned hi = g_fr->hdr;
    const HeaacSbrHeader &h = g_hdr[hi < n_hdr ? hi : n_hdr - 1];
    return g_fr->start && h.bs_interpol_freq && !g_fr->bs_coupling;
}

// w: aux block (bw, sumA, sumB, bandv) + record block (header, ONE channel); w.xlow / alpha0 / alpha1 are not used.
// xl: out, see above.  after_params(): as in hf_channel.
template <class Hook = NoHook>
__device__ __forceinline__ void hf_channel_rx(const HfWave &w, const float *g_noise /* LDS */,
                                              const HeaacSbrFrame *g_fr, const HeaacSbrHeader *g_hdr, unsigned n_hdr,
                                              const float *g_W, const float *st_in, float *st_out, int lane_in,
                                              v2f (&xl)[40], Hook after_params = Hook())
{
    const int lane = opaque(lane_in);
    HSTAMP(0);
    const int hdr_idx = g_fr->hdr < n_hdr ? g_fr->hdr : n_hdr - 1;
    uint32_t creg[2], hreg[3];
    {
        const uint32_t *cs = reinterpret_cast<const uint32_t *>(&g_fr->ch[0]);
        const uint32_t *hs_ = reinterpret_cast<const uint32_t *>(&g_hdr[hdr_idx]);
#pragma unroll
        for (int r = 0; r < 2; r++) creg[r] = lane + 64 * r < 84 ? cs[lane + 64 * r] : 0;
#pragma unroll
        for (int r = 0; r < 3; r++) hreg[r] = lane + 64 * r < 133 ? hs_[lane + 64 * r] : 0;
    }
    const int reset = g_fr->reset;
    const int kx_old = g_fr->kx_old, m_old = g_fr->m_old;
    const int k = lane;                                  // this lane's QMF band
    // ---- W[.][i][k] of this lane's band, i = 0..31, behind the previous frame's last eight slots ----
    // (the lanes from 32 on load their lower twin's column -- the same lines -- and lose it to the kx test below)
    {
        // (only the 32 lanes that have a band load: a lane's request costs the memory path whether or not its line is shared)
        const GBuf Wb(g_W), Tb(st_in);
        const int kb = opaque((lane & 31) * 8);
        const v2f zero = {0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < 40; i++) xl[i] = zero;
        if (lane < 32) {
            // slots 24..31 first: they are stored again (the next frame's tail) as soon as they are here
#pragma unroll
            for (int i = 24; i < 32; i++) xl[8 + i] = Wb.ldb2(kb, i * 64);
#pragma unroll
            for (int i = 0; i < 24; i++) xl[8 + i] = Wb.ldb2(kb, i * 64);
#pragma unroll
            for (int i = 0; i < 8; i++) xl[i] = Tb.ldb2(kb, HEAAC_SBR_WTAIL + i * 64);
        }
    }
    unsigned idxnoise = __float_as_uint(st_in[HEAAC_SBR_IDXNOISE]);
    unsigned idxsine  = __float_as_uint(st_in[HEAAC_SBR_IDXSINE]);
    const float bw_in = lane < 5 ? st_in[HEAAC_SBR_BW + lane] : 0.0f;
    {
        uint32_t *cd = reinterpret_cast<uint32_t *>(&w.c[0]);
        uint32_t *hd = reinterpret_cast<uint32_t *>(&w.h);
#pragma unroll
        for (int r = 0; r < 2; r++) if (lane + 64 * r < 84) cd[lane + 64 * r] = creg[r];
#pragma unroll
        for (int r = 0; r < 3; r++) if (lane + 64 * r < 133) hd[lane + 64 * r] = hreg[r];
    }
    after_params();
    wave_sync();
    const HeaacSbrHeader &h = w.h;
    const HeaacSbrChannel &c = w.c[0];
    const int kx = h.kx, m_max = h.m, n_q = h.n_q;
    const int m = k - kx;
    const bool in_sbr = m >= 0 && m < m_max && m < MAXM;
    const int num_env = c.bs_num_env;
    const int t0 = c.t_env[0], tL = c.t_env[num_env];
    const int h_SL = 4 * !h.bs_smoothing_mode;
    const int sidx0 = in_sbr ? reinterpret_cast<const uint8_t *>(st_in + HEAAC_SBR_SIDX)[m] : 0;
    if (reset) idxnoise = 0;                     // sbr_make_f_derived, :587-588

    HSTAMP(1);
    // ---- sbr_lf_gen (:1337-1357) ----
    {
        // new tail = W[1][24..31], as it came (the NEXT frame zeroes it against its own kx[0])
        const GBufT<0> To(st_out);
        if (lane < 32) {
#pragma unroll
            for (int j = 0; j < 8; j++) To.stb2(xl[32 + j], opaque(lane * 8), HEAAC_SBR_WTAIL + j * 64);
        }
        const v2f zero = {0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < 32; i++) xl[8 + i] = k >= kx ? zero : xl[8 + i];
#pragma unroll
        for (int i = 0; i < 8; i++) xl[i] = k >= kx_old ? zero : xl[i];
    }
    if (lane < 8) w.bw[lane] = bw_in;
    wave_sync();

    float e_orig[MAXE], q_map[MAXE], e_curr[MAXE], gain[MAXE], q_m[MAXE], s_m[MAXE];
    int sidx[MAXE];
    unsigned smap = 0;
    float kc[4] = { 0, 0, 0, 0 };
#pragma unroll
    for (int e = 0; e < MAXE; e++) { e_orig[e] = q_map[e] = e_curr[e] = gain[e] = q_m[e] = s_m[e] = 0.0f; sidx[e] = 0; }
    const int p_src = in_sbr ? (int)h.map_src[k] : 0xff;
    const bool has_src = p_src < 32;
    const int src_addr = (has_src ? p_src : lane) * 4;       // ds_bpermute address of the patch source lane
    const int t_old = c.t_env_num_env_old;
    const int i_Temp = 2 * t_old - 32 > 0 ? 2 * t_old - 32 : 0;

    {
        HSTAMP(2);
        // ---- sbr_hf_inverse_filter (:1261-1313) + autocorrelate (:1232-1255), band k in lane k ----
        // Every running sum of the reference is  s0 += a c + b d,  s1 += a d - b c  with (a, b) = x[i], (c, d) = x[i + lag],
        // lag 0 (real_sum0), 1 (real_sum1, imag_sum1) and 2 (real_sum2, imag_sum2), i = 1 .. 37 in order; the boundary
        // terms (:1244-1253) have the same form with i = 0 and i = 38.  (Plain f32 here: as packed pairs the rotated operand
        // (d, -c) needs a register pair of its own per term, and with the forty pairs of xl[] live that spills.)
        float a0r = 0.0f, a0i = 0.0f, a1r = 0.0f, a1i = 0.0f;
        {
            float r0 = 0.0f;
            v2f s1 = {0.0f, 0.0f}, s2 = {0.0f, 0.0f};
#pragma unroll
            for (int i = 1; i < 38; i++) {
                const v2f x = xl[i];
                r0 += x.x * x.x + x.y * x.y;
                { const v2f y = xl[i + 1]; s1.x += x.x * y.x + x.y * y.y; s1.y += x.x * y.y - x.y * y.x; }
                { const v2f y = xl[i + 2]; s2.x += x.x * y.x + x.y * y.y; s2.y += x.x * y.y - x.y * y.x; }
                __builtin_amdgcn_sched_barrier(0);
            }
            // sums with the i = 0 term (phi[.][1][.] of the reference) and with the i = 38 term (phi[.][0][.])
            auto edge = [](v2f s, v2f x, v2f y) {
                return v2f{s.x + x.x * y.x + x.y * y.y, s.y + x.x * y.y - x.y * y.x};
            };
            const float p210 = r0 + xl[0].x * xl[0].x + xl[0].y * xl[0].y;
            const float p100 = r0 + xl[38].x * xl[38].x + xl[38].y * xl[38].y;
            const v2f lo1 = edge(s1, xl[0], xl[1]), hi1 = edge(s1, xl[38], xl[39]);
            const v2f lo2 = edge(s2, xl[0], xl[2]);
            const float p110 = lo1.x, p111 = lo1.y, p000 = hi1.x, p001 = hi1.y, p010 = lo2.x, p011 = lo2.y;
            const float dk = p210 * p100 - (p110 * p110 + p111 * p111) / 1.000001f;
            if (dk != 0.0f) {
                const float tr = p000 * p110 - p001 * p111 - p010 * p100;
                const float ti = p000 * p111 + p001 * p110 - p011 * p100;
                a1r = tr / dk;
                a1i = ti / dk;
            }
            if (p100 != 0.0f) {
                const float tr = p000 + a1r * p110 + a1i * p111;
                const float ti = p001 + a1i * p110 - a1r * p111;
                a0r = -tr / p100;
                a0i = -ti / p100;
            }
            if (a1r * a1r + a1i * a1i >= 16.0f || a0r * a0r + a0i * a0i >= 16.0f) {
                a1r = 0; a1i = 0; a0r = 0; a0i = 0;
            }
        }
        HSTAMP(3);
        // ---- sbr_chirp (:1316-1334) ----
        if (lane < n_q) {
            const int m0 = c.bs_invf_mode[0][lane], m1 = c.bs_invf_mode[1][lane];
            float new_bw;
            if (m0 + m1 == 1) new_bw = 0.6f;
            else new_bw = m0 == 0 ? 0.0f : m0 == 1 ? 0.75f : m0 == 2 ? 0.9f : 0.98f;
            const float old = w.bw[lane];
            if (new_bw < old) new_bw = 0.75f    * new_bw + 0.25f    * old;
            else              new_bw = 0.90625f * new_bw + 0.09375f * old;
            w.bw[lane] = new_bw < 0.015625f ? 0.0f : new_bw;
        }
        wave_sync();

        HSTAMP(4);
        // ---- per-band constants of sbr_hf_gen (:1369-1386): the source band's coefficients come from its lane ----
        {
            const float s1r = lane_get(src_addr, a1r), s1i = lane_get(src_addr, a1i);
            const float s0r = lane_get(src_addr, a0r), s0i = lane_get(src_addr, a0i);
            if (has_src) {
                const int g = h.map_nq[k];
                const float b = w.bw[g < 5 ? g : 0];
                kc[0] = s1r * b * b;
                kc[1] = s1i * b * b;
                kc[2] = s0r * b;
                kc[3] = s0i * b;
            }
        }

        HSTAMP(5);
        // ---- sbr_hf_gen (:1388-1402) into the SBR lanes' own registers, and sbr_env_estimate (:1499-1526, the
        // interpolating form) over it as it is formed: one walk over X_high[k][2..39], the envelope the slot belongs to
        // being wave-uniform.  The slots sbr_x_gen fills from the PREVIOUS frame's range (i < i_Temp) keep their
        // X_low (a band that was a low band then is read from it); their Y is never used.
        {
            v2f x2 = lane_get2(src_addr, xl[0]), x1 = lane_get2(src_addr, xl[1]);
            int e = -1, next = 2 * t0 + ENV_ADJ;               // the walk enters envelope e + 1 at slot `next`
            float sum = 0.0f, recip = 0.0f;
            const int last = 2 * tL + ENV_ADJ;
#pragma unroll
            for (int j = 2; j < 40; j++) {
                const v2f x0 = lane_get2(src_addr, xl[j]);
                if ((j & 1) == 0 && j == next) {
                    // (borders are even: 2 t_env + 2)
                    if (e >= 0) {
                        const float v = sum * recip;
#pragma unroll
                        for (int q = 0; q < MAXE; q++) if (q == e) e_curr[q] = v;
                    }
                    e++;
                    sum = 0.0f;
                    if (e < num_env) {
                        recip = 0.5f / (c.t_env[e + 1] - c.t_env[e]);
                        next = 2 * c.t_env[e + 1] + ENV_ADJ;
                    } else {
                        next = 64;
                    }
                }
                v2f xh = xhigh3_pk(x2, x1, x0, kc);
                xh = has_src ? xh : v2f{0.0f, 0.0f};
                if (e >= 0 && j < last) sum += xh.x * xh.x + xh.y * xh.y;
                if (j - 2 >= 6) xl[j] = in_sbr ? xh : xl[j];
                else            xl[j] = (in_sbr && j - 2 >= i_Temp) ? xh : xl[j];
                x2 = x1; x1 = x0;
                if ((j & 1) == 1) __builtin_amdgcn_sched_barrier(0);
            }
            // the last envelope may end at slot 40 (tL = 19)
            if (e >= 0 && e < num_env) {
                const float v = sum * recip;
#pragma unroll
                for (int q = 0; q < MAXE; q++) if (q == e) e_curr[q] = v;
            }
            if (!in_sbr) {
#pragma unroll
                for (int q = 0; q < MAXE; q++) e_curr[q] = 0.0f;
            }
        }

        HSTAMP(6);
        // ---- sbr_mapping (:1451-1496) ----
        if (in_sbr) {
            const int hi = h.map_hi[k], lo = h.map_lo[k], nq = h.map_nq[k], mid = h.map_mid[k];
#pragma unroll
            for (int e = 0; e < MAXE; e++) {
                if (e < num_env) {
                    const int res = c.bs_freq_res[e + 1];
                    e_orig[e] = deq_env(w, 0, 0, e, res ? hi : lo);
                    const int kq = (c.bs_num_noise > 1) && (c.t_env[e] >= c.t_q[1]);
                    q_map[e] = deq_noise(w, 0, 0, kq, nq);
                    if (c.bs_add_harmonic_flag && mid != 0xff)
                        sidx[e] = c.bs_add_harmonic[mid] * (e >= c.e_a[1] || (sidx0 == 1));
                }
                __builtin_amdgcn_sched_barrier(0);       // one envelope at a time: forty register pairs are spoken for
            }
        }
#pragma unroll
        for (int e = 0; e < MAXE; e++) {
            if (e < num_env) {
                const unsigned long long present = __ballot(sidx[e] != 0);
                if (in_sbr) {
                    const int res = c.bs_freq_res[e + 1];
                    const uint8_t *table = res ? h.f_tablehigh : h.f_tablelow;
                    const int bi = res ? h.map_hi[k] : h.map_lo[k];
                    const int lo_k = table[bi], hi_k = table[bi + 1];
                    const unsigned long long mask = (hi_k >= 64 ? ~0ull : ((1ull << hi_k) - 1)) & ~((1ull << lo_k) - 1);
                    if (present & mask) smap |= 1u << e;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        HSTAMP(7);
        // ---- sbr_gain_calc (:1552-1605) ----
        const int lim = in_sbr ? (int)h.map_lim[k] : 0xff;
        const bool limited = lim != 0xff;
        const float limgain = h.bs_limiter_gains == 0 ? 0.70795f :
                              h.bs_limiter_gains == 1 ? 1.0f :
                              h.bs_limiter_gains == 2 ? 1.41254f : 10000000000.0f;
        const int n_lim = h.n_lim;
#pragma unroll
        for (int e = 0; e < MAXE; e++) {
            if (e < num_env && limited) {
                const int delta = !((e == c.e_a[1]) || (e == c.e_a[0]));
                const float eo = e_orig[e], qm = q_map[e], ec = e_curr[e];
                const float temp = eo / (1.0f + qm);
                q_m[e] = sqrtf(temp * qm);
                s_m[e] = sqrtf(temp * (float)sidx[e]);
                if (!((smap >> e) & 1))
                    gain[e] = sqrtf(eo / ((1.0f + ec) * (1.0f + qm * (float)delta)));
                else
                    gain[e] = sqrtf(eo * qm / ((1.0f + ec) * (1.0f + qm)));
                w.sumA[e][m] = eo;
                w.sumB[e][m] = ec;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        wave_sync();
        for (int t = lane; t < num_env * n_lim; t += WAVE) {
            const int e = t / n_lim, kk = t - e * n_lim;
            const int ma = h.f_tablelim[kk] - kx, mb = h.f_tablelim[kk + 1] - kx;
            float sum0 = 0.0f, sum1 = 0.0f;
            for (int mm = ma; mm < mb; mm++) {
                sum0 += w.sumA[e][mm];
                sum1 += w.sumB[e][mm];
            }
            float gain_max = limgain * sqrtf((1.1920928955078125e-7f + sum0) / (1.1920928955078125e-7f + sum1));
            gain_max = FFMIN_(100000.0f, gain_max);
            w.bandv[e][kk] = gain_max;
        }
        wave_sync();
#pragma unroll
        for (int e = 0; e < MAXE; e++) {
            if (e < num_env && limited) {
                const int delta = !((e == c.e_a[1]) || (e == c.e_a[0]));
                const float gain_max = w.bandv[e][lim];
                const float q_m_max = q_m[e] * gain_max / gain[e];
                q_m[e]  = FFMIN_(q_m[e], q_m_max);
                gain[e] = FFMIN_(gain[e], gain_max);
                w.sumB[e][m] = e_curr[e] * gain[e] * gain[e]
                               + s_m[e] * s_m[e]
                               + (float)(delta && !s_m[e]) * q_m[e] * q_m[e];
            }
        }
        wave_sync();
        for (int t = lane; t < num_env * n_lim; t += WAVE) {
            const int e = t / n_lim, kk = t - e * n_lim;
            const int ma = h.f_tablelim[kk] - kx, mb = h.f_tablelim[kk + 1] - kx;
            float sum0 = 0.0f, sum1 = 0.0f;
            for (int mm = ma; mm < mb; mm++) {
                sum0 += w.sumA[e][mm];
                sum1 += w.sumB[e][mm];
            }
            float gain_boost = sqrtf((1.1920928955078125e-7f + sum0) / (1.1920928955078125e-7f + sum1));
            gain_boost = (float)(1.584893192 > (double)gain_boost ? (double)gain_boost : 1.584893192);
            w.bandv[e][kk] = gain_boost;
        }
        wave_sync();
#pragma unroll
        for (int e = 0; e < MAXE; e++) {
            if (e < num_env && limited) {
                const float gain_boost = w.bandv[e][lim];
                gain[e] *= gain_boost;
                q_m[e]  *= gain_boost;
                s_m[e]  *= gain_boost;
            }
        }
    }

    // The limiter sums are done with: their LDS now keeps this lane's gain / q_m / s_m of every envelope (fifteen
    // registers through the 38-slot loop otherwise); a new envelope reads its three values back.
    float (*genv)[3][64] = reinterpret_cast<float (*)[3][64]>(w.sumA);
    const float gain_first = gain[0], qm_first = q_m[0];
    int sidx_last = 0;
#pragma unroll
    for (int e = 0; e < MAXE; e++) {
        if (e == num_env - 1) sidx_last = sidx[e];
        genv[e][0][lane] = gain[e]; genv[e][1][lane] = q_m[e]; genv[e][2][lane] = s_m[e];
    }
    wave_sync();
    HSTAMP(8);
    // ---- sbr_hf_assemble (:1608-1714) fused with sbr_x_gen (:1412-1446), in place over xl[i + 2] ----
    const float *ytail_in = st_in + HEAAC_SBR_YTAIL;
    float *ytail_out = st_out + HEAAC_SBR_YTAIL;
    {
        const bool hf = in_sbr;
        const float hs[5] = { 0.33333333333333f, 0.30150283239582f, 0.21816949906249f,
                              0.11516383427084f, 0.03183050093751f };
        const int phi_sign0 = (1 - 2 * (kx & 1)) * ((m & 1) ? -1 : 1);
        v2f gq[5];
#pragma unroll
        for (int j = 0; j < 5; j++) gq[j] = v2f{0.0f, 0.0f};
        int e = 0, next_border = 2 * c.t_env[1];
        float g_e = gain_first, q_e = qm_first, s_e = genv[0][2][lane];
        bool plain = (0 == c.e_a[0]) || (0 == c.e_a[1]);
        const unsigned noise0 = idxnoise + (unsigned)(m + 1) - (unsigned)(2 * t0) * (unsigned)m_max;
        v2f ph_cur;
        {
            const int isine = idxsine & 3;
            const int phi_re = isine == 0 ? 1 : isine == 2 ? -1 : 0;
            const int phi_im = isine == 1 ? 1 : isine == 3 ? -1 : 0;
            ph_cur = v2f{(float)phi_re, (float)(phi_im * phi_sign0)};
        }
        const v2f ph_rot = v2f{(float)-phi_sign0, (float)phi_sign0};
        v2f nq[4];
#pragma unroll
        for (int i = 0; i < 38; i++) {
            if ((i & 3) == 0) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (i + j < 38) {
                        const unsigned in = (noise0 + (unsigned)(i + j) * (unsigned)m_max) & 0x1ff;
                        nq[j] = v2f{g_noise[2 * in], g_noise[2 * in + 1]};
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if ((i & 1) == 0 && i <= 6 && i == 2 * t0 && h_SL) {
                // the four g_temp / q_temp history rows of band m (:1630-1639; after a reset: the first envelope's
                // values).  Fetched here, where smoothing frames use them: held from the top of the stage they cost
                // every frame eight registers through the walk and the limiter
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float gh = in_sbr ? st_in[HEAAC_SBR_GTAIL + j * MAXM + m] : 0.0f;
                    float qh = in_sbr ? st_in[HEAAC_SBR_QTAIL + j * MAXM + m] : 0.0f;
                    if (reset) { gh = gain_first; qh = qm_first; }
                    gq[(i + j) % 5] = v2f{gh, qh};
                }
            }
            if ((i & 1) == 0 && i > 0 && i == next_border && e + 1 < num_env) {
                e++;
                next_border = 2 * c.t_env[e + 1];
                g_e = genv[e][0][lane]; q_e = genv[e][1][lane]; s_e = genv[e][2][lane];
                plain = (e == c.e_a[0]) || (e == c.e_a[1]);
            }
            v2f Y = v2f{0.0f, 0.0f};
            const bool in_time = i >= 2 * t0 && i < 2 * tL;           // uniform
            const bool have_y = hf && in_time;
            if (in_time) {
                // X_high[k][i + ENV_ADJ], formed above (the slots below i_Temp hold X_low instead: their Y goes nowhere)
                const v2f xh = xl[i + ENV_ADJ];
                gq[(i + 4) % 5] = v2f{g_e, q_e};
                v2f f = v2f{g_e, q_e};
                if (h_SL && !plain) {
                    v2f a = v2f{0.0f, 0.0f};
#pragma unroll
                    for (int j = 0; j < 5; j++) a = a + gq[(i + 4 - j) % 5] * bc(hs[j]);
                    f = a;
                }
                Y = xh * bc(f.x);
                const v2f ph = ph_cur;
                ph_cur = ph_rot * swp(ph_cur) + v2f{0.0f, 0.0f};
                const v2f y_sine = Y + bc(s_e) * ph;
                if (!plain) {
                    const v2f y_noise = Y + bc(f.y) * nq[i & 3];
                    Y = s_e != 0.0f ? y_sine : y_noise;
                } else {
                    Y = y_sine;
                }
                Y = hf ? Y : v2f{0.0f, 0.0f};
            }
            // ytail: Y[1][32..37]
            if (i >= 32) {
                const int o = ((i - 32) * 64 + k) * 2;
                if (have_y) { ytail_out[o] = Y.x; ytail_out[o + 1] = Y.y; }
                else if (ytail_out != ytail_in) { ytail_out[o] = ytail_in[o]; ytail_out[o + 1] = ytail_in[o + 1]; }
            }
            // x_gen: the low bands keep their X_low, which is where it already stands
            const v2f zero = {0.0f, 0.0f};
            if (i < 6 && i < i_Temp) {
                // (the previous frame's Y tail is fetched where it is used: frames whose first slots follow the old range are
                // the exception, and six register pairs held for them through the whole stage cost every frame)
                const bool lo = k < kx_old, hi = !lo && k < kx_old + m_old;
                v2f yt = zero;
                if (hi) yt = v2f{ytail_in[(i * 64 + k) * 2], ytail_in[(i * 64 + k) * 2 + 1]};
                xl[i + 2] = (lo && k < 32) ? xl[i + 2] : yt;
            } else {
                const bool lo32 = k < kx && k < 32, hi = k >= kx && k < kx + m_max && i < 32;
                xl[i + 2] = lo32 ? xl[i + 2] : hi ? Y : zero;
            }
        }
    }

    HSTAMP(9);
    // ---- remaining state ----
    {
        if (lane < 5) st_out[HEAAC_SBR_BW + lane] = w.bw[lane];
        if (lane == 0) {
            const unsigned slots = 2 * (tL - t0);
            st_out[HEAAC_SBR_IDXNOISE] = __uint_as_float((idxnoise + slots * m_max) & 0x1ff);
            st_out[HEAAC_SBR_IDXSINE]  = __uint_as_float((idxsine + slots) & 3);
        }
        {
            const int v = sidx_last;
            const int src_lane = kx + lane;
            const int byte = __shfl(v, src_lane < 64 ? src_lane : 0);
            const int valid = lane < MAXM && src_lane < 64 && lane < m_max;
            const int b0 = valid ? (byte & 0xff) : 0;
            const int p0 = __shfl(b0, (lane & 15) * 4 + 0 < 64 ? (lane & 15) * 4 + 0 : 0);
            const int p1 = __shfl(b0, (lane & 15) * 4 + 1 < 64 ? (lane & 15) * 4 + 1 : 0);
            const int p2 = __shfl(b0, (lane & 15) * 4 + 2 < 64 ? (lane & 15) * 4 + 2 : 0);
            const int p3 = __shfl(b0, (lane & 15) * 4 + 3 < 64 ? (lane & 15) * 4 + 3 : 0);
            if (lane < 12)
                reinterpret_cast<uint32_t *>(st_out + HEAAC_SBR_SIDX)[lane] =
                    (uint32_t)p0 | ((uint32_t)p1 << 8) | ((uint32_t)p2 << 16) | ((uint32_t)p3 << 24);
        }
        if (h_SL) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int slot = 2 * tL - 4 + j;
                int ee = 0;
                for (int q = 1; q < num_env; q++)
                    if (slot >= 2 * c.t_env[q]) ee = q;
                const float g = genv[ee][0][lane], q = genv[ee][1][lane];
                if (m >= 0 && m < MAXM) {
                    st_out[HEAAC_SBR_GTAIL + j * MAXM + m] = in_sbr ? g : 0.0f;
                    st_out[HEAAC_SBR_QTAIL + j * MAXM + m] = in_sbr ? q : 0.0f;
                }
            }
            if (lane < MAXM && kx + lane >= 64) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    st_out[HEAAC_SBR_GTAIL + j * MAXM + lane] = 0.0f;
                    st_out[HEAAC_SBR_QTAIL + j * MAXM + lane] = 0.0f;
                }
            }
        } else if (st_out != st_in) {
            for (int t = lane; t < 4 * MAXM; t += WAVE) {
                st_out[HEAAC_SBR_GTAIL + t] = st_in[HEAAC_SBR_GTAIL + t];
                st_out[HEAAC_SBR_QTAIL + t] = st_in[HEAAC_SBR_QTAIL + t];
            }
        }
    }
    if (lane == 0 && st_out != st_in) st_out[HEAAC_SBR_PAD] = st_in[HEAAC_SBR_PAD];
    wave_sync();
    HSTAMP(10);
}
