"""What-if variants of the HF stage (k_hf.h; invalid audio, valid timing): name = hf_<part>
  noenv   sbr_env_estimate skipped      noinv  inverse filter / autocorrelation skipped
  nogain  sbr_gain_calc's limiter rounds skipped      nox   the slot loop's X_high / smoothing arithmetic skipped"""
import sys
d, name = sys.argv[1], sys.argv[2]
p = d + '/k_hf.h'
s = open(p).read()
part = name.split('_')[1]
def rep(a, b):
    global s
    assert a in s, a[:60]
    s = s.replace(a, b, 1)
if part == 'noenv':
    rep('        if (h.bs_interpol_freq) {\n            if (in_sbr) {', '        if (h.bs_interpol_freq) {\n            if (in_sbr && lane == 99) {')
elif part == 'noinv':
    rep('        if (lane < h.k0 && lane < 32) {\n            // the whole row first', '        if (lane < h.k0 && lane < 32 && lane == 99) {\n            // the whole row first')
elif part == 'nogain':
    rep('        for (int t = lane; t < num_env * n_lim; t += WAVE) {\n            const int e = t / n_lim, kk = t - e * n_lim;\n            const int ma = h.f_tablelim[kk] - kx, mb = h.f_tablelim[kk + 1] - kx;\n            float sum0 = 0.0f, sum1 = 0.0f;\n            for (int mm = ma; mm < mb; mm++) {\n                sum0 += w.sumA[e][mm];\n                sum1 += w.sumB[e][mm];\n            }\n            float gain_max',
        '        for (int t = lane; t < num_env * n_lim; t += WAVE) {\n            const int e = t / n_lim, kk = t - e * n_lim;\n            const int ma = h.f_tablelim[kk] - kx, mb = ma + 1;\n            float sum0 = 0.0f, sum1 = 0.0f;\n            for (int mm = ma; mm < mb; mm++) {\n                sum0 += w.sumA[e][mm];\n                sum1 += w.sumB[e][mm];\n            }\n            float gain_max')
elif part == 'nox':
    rep('                v2f xh = xhigh3_pk(x2, x1, x0, kc);', '                v2f xh = x0;')
else:
    raise SystemExit('unknown ' + part)
open(p, 'w').write(s)
