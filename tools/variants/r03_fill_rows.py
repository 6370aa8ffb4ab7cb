"""What-if: the fused kernel also writes (zeros into) the bins of the PS delay-line and all-pass rows that the 20-band
layout does not use, so that every row of those blocks is written whole and the blocks become contiguous runs of full
128-byte lines.  Invalid state for streams that switch layouts; valid timing: does partial-line write-back cost?"""
import sys
p = sys.argv[1] + '/k_psf.h'
s = open(p).read()
old = "    // bands that exist in the state record but not in this layout / all-pass set\n"
assert old in s
new = '''    if constexpr (FUSED) {
        const v2f z2 = {0.0f, 0.0f};
        if (lane < 20) {
#pragma unroll
            for (int j = 0; j < 14; j++) SO.stb2(z2, opaque((71 + lane) * 8), HEAAC_PS_DELAY + j * 91 * 2);
#pragma unroll
            for (int j = 0; j < 15; j++) SO.stb2(z2, opaque((30 + lane) * 8), HEAAC_PS_APDELAY + j * 50 * 2);
        }
    }
'''
s = s.replace(old, new + old)
open(p, 'w').write(s)
