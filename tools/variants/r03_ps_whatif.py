"""What-ifs of the baseline PS stage, measured on the UNFUSED kernel k_ps<false, 8> (-DHEAAC_TUNING build, HEAAC_HE_UNFUSED=1;
invalid audio): name = pw_<part>
  nox    no X stores in the slot loop       nost   no state stores (delay tail, all-pass rings)
  noap   all-pass chain skipped             nomix  the mixing arithmetic skipped (outputs = inputs)
  noloop the whole slot loop (pass 1) skipped"""
import sys
d, name = sys.argv[1], sys.argv[2]
p = d + '/k_psf.h'
s = open(p).read()
part = name.split('_')[1]
def rep(a, b):
    global s
    assert a in s, a[:70]
    s = s.replace(a, b, 1)
if part == 'nox':
    rep('        if (X_BANDS == 64 || q < X_BANDS) {\n            const int qb = opaque(qs4);', '        if (q < 0) {\n            const int qb = opaque(qs4);')
elif part == 'nost':
    rep('        SO.stb2(v, kv, HEAAC_PS_DELAY + j * dl_stride);\n    }', '        if (q < 0) SO.stb2(v, kv, HEAAC_PS_DELAY + j * dl_stride);\n    }')
    rep('                SO.stb2(ring[m][(27 + j) % 5], kv, HEAAC_PS_APDELAY + (m * 5 + j) * ap_stride);', '                if (q < 0) SO.stb2(ring[m][(27 + j) % 5], kv, HEAAC_PS_APDELAY + (m * 5 + j) * ap_stride);')
elif part == 'noap':
    rep('#pragma unroll\n            for (int m = 0; m < 3; m++) {\n                const v2f a = bc(ag[m]) * x;', '#pragma unroll\n            for (int m = 0; m < 0; m++) {\n                const v2f a = bc(ag[m]) * x;')
elif part == 'nomix':
    rep('        v2f lv = bc(hA.x) * sv + bc(hB.x) * rv;\n        v2f rr = bc(hA.y) * sv + bc(hB.y) * rv;', '        v2f lv = sv + rv;\n        v2f rr = sv - rv;')
elif part == 'noloop':
    rep('            if (aligned8 && x_bands == 48)', '            if (lane_in == 999) {} else if (aligned8 && x_bands == 777) {} if (false)')
else:
    raise SystemExit('unknown ' + part)
open(p, 'w').write(s)
