"""What-if on the top-aware X hand-over: take the +0 proof as given for every frame (x_bands = top16 always)."""
import sys
p = sys.argv[1] + '/k_psf.h'
s = open(p).read()
old = "            if (__builtin_amdgcn_readfirstlane(__ballot(bad) == 0ull)) x_bands = top16;"
assert old in s
s = s.replace(old, "            (void)bad; x_bands = top16;")
open(p, 'w').write(s)
