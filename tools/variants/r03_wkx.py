"""What-if: W rows only up to band 16 (the headline header's kx = 13, rounded up to one 128-byte line): k_core_ana does
not store the bands above, the HF stage does not load them.  Bounds what a kx-aware W hand-over could gain."""
import sys
d = sys.argv[1]
p = d + '/k_he.hip'; s = open(p).read()
old = '''                Wo[t] = uu[(t >> 6) * 65 + (t & 63)];'''
assert old in s
s = s.replace(old, '''                if ((t & 63) < 32) Wo[t] = uu[(t >> 6) * 65 + (t & 63)];''')
open(p, 'w').write(s)
p = d + '/k_hf.h'; s = open(p).read()
old = '''        for (int r = 0; r < 16; r++) wreg[r] = W2[lane + 64 * r];'''
assert old in s
s = s.replace(old, '''        for (int r = 0; r < 16; r++) wreg[r] = (lane & 31) < 16 ? W2[lane + 64 * r] : make_float2(0.0f, 0.0f);''')
open(p, 'w').write(s)
