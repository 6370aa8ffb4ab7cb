"""What-if: k_synth without the 64 IMDCTs (inputs passed through)."""
import sys
p = sys.argv[1] + '/k_he.hip'
s = open(p).read()
a = '''    imdct128_reg([&](int j) -> float {
                     return (j & 1) ? __uint_as_float(__float_as_uint(x[j]) ^ flip) : x[j];
                 }, o, S.rot, S.c16, S.c32);
    SSTAMP(2);'''
assert a in s
s = s.replace(a, '''    for (int j = 0; j < 64; j++) o[j] = (j & 1) ? __uint_as_float(__float_as_uint(x[j]) ^ flip) : x[j];
    SSTAMP(2);''')
open(p, 'w').write(s)
