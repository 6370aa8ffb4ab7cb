"""What-if for HE-AACv1: k_hfadj stores only the 32 slots k_synth reads (a: all 64 bands, b: bands below 48)."""
import sys
d, name = sys.argv[1], sys.argv[2]
p = d + '/k_he.hip'; s = open(p).read()
old = '''                       __builtin_nontemporal_store(re, X0 + i * 64 + lane);
                       __builtin_nontemporal_store(im, X1 + i * 64 + lane);'''
assert old in s
cond = 'i < 32' if name.endswith('a') else 'i < 32 && lane < 48'
s = s.replace(old, 'if (%s) {\n' % cond + old + '\n}')
open(p, 'w').write(s)
