"""What-if: k_core_ana's long-window values as lane constants instead of loads from the table blob (bounds what prefetching
them could buy)."""
import sys
p = sys.argv[1] + '/k_core2.h'
s = open(p).read()
old = '''            const float wi = lwindow_prev[p], wj = lwindow_prev[1023 - p];'''
assert old in s
s = s.replace(old, '''            const float wi = (float)p * 1e-3f, wj = 1.0f - (float)p * 1e-3f; (void)lwindow_prev;''')
open(p, 'w').write(s)
