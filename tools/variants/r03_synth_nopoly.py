"""What-if: k_synth without the polyphase sum (emits one v value per output instead of the 10-tap sum)."""
import sys
p = sys.argv[1] + '/k_he.hip'
s = open(p).read()
a = s.index('        v2f acc = v2f{va[0], vb[0]} * bc(wt[0]) + v2f{0.0f, 0.0f};')
b = s.index('        if (scale_and_bias) acc = acc * bc(scale) + bc(bias);')
s = s[:a] + '        v2f acc = v2f{va[0], vb[0]} * bc(wt[0]) + v2f{0.0f, 0.0f};\n' + s[b:]
open(p, 'w').write(s)
