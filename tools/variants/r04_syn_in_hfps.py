"""What-if of VERDICT r03 #2 ("day-one what-if"): the fused HF + PS kernel with the X row stores of the slot loop replaced by
the synthesis bank's WORK on dummy data -- does the wave absorb the 64-point transforms and the window sums?  Invalid audio,
valid timing of k_hfps (k_synth still runs behind it on garbage).  name = syn_<part>:
  stage   the slot loop writes (L, R) of every slot to an LDS row instead of the two global row stores; nothing else
  full    ... and behind the PS stage the wave runs, per channel, what k_synth runs on chip: 64 staged-row reads (b128),
          64 register IMDCT-128, the DPP butterfly into v rows, the 10-tap polyphase sum (LDS reads, packed f32), and stores
          2048 floats of "PCM" per channel (into the X workspace: 16 KiB per frame instead of the 27 KiB of X rows).
          The LDS it works in is the wave's own (X_low / scratch block, 2 048 words, indices wrapped): the real thing
          needs 17 - 21 KB more per wave, which is why this is a what-if.
"""
import sys
d, name = sys.argv[1], sys.argv[2]
part = name.split('_')[1]

p = d + '/k_psf.h'
s = open(p).read()
old = '''            const int qb = opaque(qs8);
            X.stb2(lv, qb, n * 128);
            X.stb2(rr, qb, XC + n * 128);'''
assert old in s
s = s.replace(old, '''            if constexpr (W::IS_GENERAL) {
                const int qb = opaque(qs8);
                X.stb2(lv, qb, n * 128);
                X.stb2(rr, qb, XC + n * 128);
            } else {
                // what-if: the row goes to an LDS stage (the hybrid analysis' input block is dead here: 1 056 bytes)
                float *stage = &w.inb[0][0][0];
                *reinterpret_cast<v2f *>(stage + 2 * (q & 63)) = lv;
                *reinterpret_cast<v2f *>(stage + 128 + 2 * (q & 63)) = rr;
            }''')
open(p, 'w').write(s)

if part == 'full':
    p = d + '/k_ps.hip'
    s = open(p).read()
    fn = r'''
// ---- what-if: the synthesis bank's on-chip work (k_he.hip: syn_rows + syn_poly) on dummy data, in 2 048 words of LDS ----
__device__ __forceinline__ float wi_xor1(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
}
__device__ __forceinline__ void synth_whatif(float *vb, const float *__restrict__ g_tab, float *g_out, int lane)
{
#define VB(i) vb[(i) & 2047]
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (int ch = 0; ch < 2; ch++) {
        // the staged image -> this lane's row (16 b128 reads)
        float x[64];
        const int mine = (lane & 1) * 32 + (lane >> 1);
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const f4 t = *reinterpret_cast<const f4 *>(&VB(((mine * 68) & ~3) + 4 * q));
            x[4 * q] = t.x; x[4 * q + 1] = t.y; x[4 * q + 2] = t.z; x[4 * q + 3] = t.w;
        }
        wave_sync();
        // history rows
#pragma unroll
        for (int r = 0; r < 18; r++) VB((32 + (r >> 1)) * 129 + (r & 1) * 64 + lane) = x[r];
        const int i = lane >> 1, part = lane & 1;
        const unsigned flip = (unsigned)part << 31;
        float o[64];
        imdct128_reg([&](int j) -> float {
                         return (j & 1) ? __uint_as_float(__float_as_uint(x[j]) ^ flip) : x[j];
                     }, o, g_tab + TB_ROT128S, g_tab + TB_COS16, g_tab + TB_COS32);
        const unsigned neg = (unsigned)(part ^ 1) << 31;
        const int vs = (31 - i) * 129 + 64 * part;
#pragma unroll
        for (int n = 0; n < 64; n++) {
            const float mv = __uint_as_float(__float_as_uint(o[63 - n]) ^ neg);
            VB(vs + n) = wi_xor1(o[n]) + mv;
        }
        wave_sync();
        float wt[10];
#pragma unroll
        for (int j = 0; j < 10; j++) wt[j] = g_tab[TB_QMF_US + 64 * j + lane];
        float *o_ch = g_out + ch * 2048;
        for (int s2 = 0; s2 < 32; s2 += 2) {
            const int va = (31 - s2) * 129 + lane, vbb = va - 129;
            v2f acc = v2f{VB(va), VB(vbb)} * bc(wt[0]) + v2f{0.0f, 0.0f};
#pragma unroll
            for (int j = 1; j < 10; j++)
                acc = v2f{VB(va + j * 129 + (j & 1) * 64), VB(vbb + j * 129 + (j & 1) * 64)} * bc(wt[j]) + acc;
            acc = acc * bc(1.0000001f) + bc(385.0f);
            __builtin_nontemporal_store(acc.x, o_ch + 64 * s2 + lane);
            __builtin_nontemporal_store(acc.y, o_ch + 64 * (s2 + 1) + lane);
        }
        wave_sync();
    }
#undef VB
}
'''
    s = s.replace('#define HFPS_WAVES 8\n', fn + '\n#define HFPS_WAVES 8\n', 1)
    old = '''            ps_frame<false, true>(W, g_tab, &g_ps[f], top, st_in + off_ps, st_out + off_ps, Xf, lane, wave, col,
                                  prefetch_next, g_xtop + 2 * f, x_zero_above);
        }
        feed.advance();'''
    assert old in s
    s = s.replace(old, '''            ps_frame<false, true>(W, g_tab, &g_ps[f], top, st_in + off_ps, st_out + off_ps, Xf, lane, wave, col,
                                  prefetch_next, g_xtop + 2 * f, x_zero_above);
            synth_whatif(s_a[wave], g_tab, Xf, lane);
        }
        feed.advance();''', 1)
    open(p, 'w').write(s)
