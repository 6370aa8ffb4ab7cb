"""What-if variants of k_hfps's L2 touch prefetch (round 3): where in the PS slot loop the next frame's
W / SBR state / PS state are touched.  usage: VARIANT_EDIT=tools/variants/r03_touch.py tools/build_variants.sh t28s ""
name = t<slot><s|a>: s = W + SBR state, a = + PS delay lines."""
import re, sys
d, name = sys.argv[1], sys.argv[2]
m = re.match(r't(\d+)([sa])', name)
slot, what = int(m.group(1)), m.group(2)
p = d + '/k_ps.hip'
s = open(p).read()
old = '''                l2_touch(&g_ps[f1], sizeof(HeaacPsFrame), lane, sink);
            }
'''
new = old + '''            if (n == %d && f1 < n_frames) {
                l2_touch(g_W + f1 * 2048, 2048 * 4, lane, sink);
                l2_touch(g_state_in + f1 * state_words + off_sbr, HEAAC_ST_SBR * 4, lane, sink);
%s            }
''' % (slot, '                l2_touch(g_state_in + f1 * state_words + off_ps, HEAAC_ST_PS * 4, lane, sink);\n' if what == 'a' else '')
assert old in s
open(p, 'w').write(s.replace(old, new))
