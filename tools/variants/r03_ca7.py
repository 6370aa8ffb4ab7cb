"""What-if: k_core_ana with seven waves per CU (channel 1's analysis input shortened so that it runs into the next
array: wrong audio, the same instruction stream and LDS traffic).  Bounds what freeing 15 KB of table space in LDS
(long windows and rotation tables read from global memory instead) could buy."""
import sys
p = sys.argv[1] + '/k_he.hip'
s = open(p).read()
for a, b in (('#define CA_WAVES 6', '#define CA_WAVES 7'), ('    float x1[1312];', '    float x1[776];')):
    assert a in s
    s = s.replace(a, b, 1)
open(p, 'w').write(s)
