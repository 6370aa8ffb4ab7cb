"""What-if: pace the slot loop's X stores with s_sleep (name = pace<N>: s_sleep N after every slot's stores)."""
import re, sys
d, name = sys.argv[1], sys.argv[2]
n = int(re.match(r'pace(\d+)', name).group(1))
p = d + '/k_psf.h'
s = open(p).read()
old = '''            X.stb2(rr, qb, XC + n * 128);
        }'''
assert old in s
s = s.replace(old, '''            X.stb2(rr, qb, XC + n * 128);
            __builtin_amdgcn_s_sleep(%d);
        }''' % n)
open(p, 'w').write(s)
