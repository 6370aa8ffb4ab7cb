"""Cache-policy variants of the fused kernel's stores: name = aux<X rows>_<state>  (auxiliary bits of the buffer store:
1 = sc0, 2 = nt, 16 = sc1)."""
import re, sys
d, name = sys.argv[1], sys.argv[2]
m = re.match(r'aux(\d+)_(\d+)', name)
xr, so = int(m.group(1)), int(m.group(2))
p = d + '/k_psf.h'
s = open(p).read()
a = 'typedef GBufT<2> GBufSO;'
b = 'typedef GBufT<0> GBufXR;'
assert a in s and b in s
s = s.replace(a, 'typedef GBufT<%d> GBufSO;' % so).replace(b, 'typedef GBufT<%d> GBufXR;' % xr)
open(p, 'w').write(s)
