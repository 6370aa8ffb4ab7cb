"""More what-ifs of the HF stage (invalid audio): name = hg_<part>
  nomap   sbr_mapping skipped            nogain2  the whole of sbr_gain_calc skipped
  nosmooth  the 5-tap gain smoothing skipped    nonoise  noise / sinusoid addition skipped
  nostore  k_hfadj's X stores skipped    nolf  sbr_lf_gen's LDS stores skipped"""
import sys
d, name = sys.argv[1], sys.argv[2]
part = name.split('_')[1]
def edit(path, a, b):
    s = open(path).read()
    assert a in s, a[:60]
    open(path, 'w').write(s.replace(a, b, 1))
hf = d + '/k_hf.h'
if part == 'nomap':
    edit(hf, '        if (in_sbr) {\n            const int hi = h.map_hi[k], lo = h.map_lo[k], nq = h.map_nq[k], mid = h.map_mid[k];',
             '        if (in_sbr && lane == 99) {\n            const int hi = h.map_hi[k], lo = h.map_lo[k], nq = h.map_nq[k], mid = h.map_mid[k];')
elif part == 'nogain2':
    edit(hf, '        const int lim = in_sbr ? (int)h.map_lim[k] : 0xff;\n        const bool limited = lim != 0xff;',
             '        const int lim = in_sbr ? (int)h.map_lim[k] : 0xff;\n        const bool limited = lim != 0xff && lane == 99;')
elif part == 'nosmooth':
    edit(hf, '                if (h_SL && !plain) {\n                    v2f a = v2f{0.0f, 0.0f};', '                if (h_SL && !plain && lane == 99) {\n                    v2f a = v2f{0.0f, 0.0f};')
elif part == 'nonoise':
    edit(hf, '                if (!plain) {\n                    // sbr_noise_table', '                if (!plain && i == 99) {\n                    // sbr_noise_table')
elif part == 'nostore':
    edit(d + '/k_he.hip', '                       __builtin_nontemporal_store(re, X0 + i * 64 + lane);\n                       __builtin_nontemporal_store(im, X1 + i * 64 + lane);',
         '                       if (re == 1.2345e-30f) { __builtin_nontemporal_store(re, X0 + i * 64 + lane);\n                       __builtin_nontemporal_store(im, X1 + i * 64 + lane); }')
elif part == 'nolf':
    edit(hf, '            float *d = w.xlow + kk * XL_STRIDE + 2 * (i + 8);\n            d[0] = v.x; d[1] = v.y;', '            float *d = w.xlow + kk * XL_STRIDE + 2 * (i + 8);\n            if (r == 0) { d[0] = v.x; d[1] = v.y; }')
else:
    raise SystemExit('unknown ' + part)
