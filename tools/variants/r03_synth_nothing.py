"""What-if: k_synth with neither global loads nor global stores (frame loop, queue, LDS work and arithmetic only)."""
import sys, re
p = sys.argv[1] + '/k_he.hip'
s = open(p).read()
a = s.index('__device__ __forceinline__ void syn_load(')
b = s.index('template <class SL>\n__device__ __forceinline__ void syn_rows(')
s = s[:a] + '''__device__ __forceinline__ void syn_load(const float *X0, const float *X1, const float *v_in, int lane, SynIn &d)
{
    const float c = (float)lane * 1e-6f + (float)((size_t)X0 & 255) * 1e-7f;
#pragma unroll
    for (int q = 0; q < 64; q++) d.x[q] = c + q * 1e-8f;
#pragma unroll
    for (int r = 0; r < 18; r++) d.h[r] = c - r * 1e-8f;
}

''' + s[b:]
s = s.replace('__device__ __forceinline__ void syn_st(T *p, T v) { __builtin_nontemporal_store(v, p); }',
              '__device__ __forceinline__ void syn_st(T *p, T v) { if (v == (T)123) __builtin_nontemporal_store(v, p); }')
open(p, 'w').write(s)
