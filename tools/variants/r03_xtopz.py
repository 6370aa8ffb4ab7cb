"""What-if: k_synth takes the X bands above 48 from ONE shared record (frame 0's, cache resident) instead of its own
frame's -- what a `top`-aware load from a page of zeros would cost, without exec-mask branches around the loads."""
import sys
d = sys.argv[1]
p = d + '/k_he.hip'; s = open(p).read()
old = '''        const float *X0 = g_X + (f * 2 + ch) * (2 * 38 * 64);
        syn_load(X0,'''
assert old in s
s = s.replace(old, '''        const float *X0 = g_X + (((lane & 15) * 4 < 48 ? f : 0ull) * 2 + ch) * (2 * 38 * 64);
        syn_load(X0,''')
open(p, 'w').write(s)
