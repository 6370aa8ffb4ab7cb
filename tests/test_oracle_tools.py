"""Domain properties of the oracle's spectral tools (the reference holds no vectors for them)."""
import importlib

import numpy as np


def _synth():
    import __graft_entry__ as g
    return importlib.import_module(g.PKG_NAME + ".synth")


def _long_frame(pkg, n=1):
    t = np.zeros(n, pkg.TOOLS_FRAME_DT)
    synth = _synth()
    for fr in t:
        for c in range(2):
            ics = fr["ch"][c]["ics"]
            ics["num_windows"] = 1; ics["num_window_groups"] = 1; ics["group_len"][0] = 1
            ics["num_swb"] = 49; ics["max_sfb"] = 49; ics["tns_max_bands"] = 40
            ics["swb_offset"][:50] = synth.SWB_1024_48
            fr["ch"][c]["band_type"][:49] = 1
    return t


def test_mid_side_is_a_butterfly(pkg, oracle):
    t = _long_frame(pkg)
    t["common_window"] = 1; t["ms_present"] = 1
    t["ms_mask"][0][:49:2] = 1                       # every other band
    rng = np.random.default_rng(0)
    c = rng.integers(-1000, 1000, (1, 2, 1024)).astype(np.float32)
    out = oracle.spectral_tools_batch(2, c, t)
    off = _synth().SWB_1024_48
    for b in range(49):
        sl = slice(off[b], off[b + 1])
        if b % 2 == 0:
            assert np.array_equal(out[0, 0, sl], c[0, 0, sl] + c[0, 1, sl])
            assert np.array_equal(out[0, 1, sl], c[0, 0, sl] - c[0, 1, sl])
        else:
            assert np.array_equal(out[0, :, sl], c[0, :, sl])
    # applying it twice doubles (exact on the integer grid)
    out2 = oracle.spectral_tools_batch(2, out, t)
    assert np.array_equal(out2[0, :, : off[1]], 2 * c[0, :, : off[1]])


def test_intensity_bands_are_scaled_copies(pkg, oracle):
    t = _long_frame(pkg)
    t["ch"][0][1]["band_type"][10] = 15              # in phase
    t["ch"][0][1]["band_type"][11] = 14              # out of phase
    t["ch"][0][1]["sf"][10] = 0.5; t["ch"][0][1]["sf"][11] = 0.25
    rng = np.random.default_rng(1)
    c = rng.standard_normal((1, 2, 1024)).astype(np.float32)
    out = oracle.spectral_tools_batch(2, c, t)
    off = _synth().SWB_1024_48
    assert np.array_equal(out[0, 1, off[10]:off[11]], np.float32(0.5) * c[0, 0, off[10]:off[11]])
    assert np.array_equal(out[0, 1, off[11]:off[12]], np.float32(-0.25) * c[0, 0, off[11]:off[12]])
    keep = np.ones(1024, bool); keep[off[10]:off[12]] = False
    assert np.array_equal(out[0, 1, keep], c[0, 1, keep]) and np.array_equal(out[0, 0], c[0, 0])
    # with ms_present the mask flips the sign (aacdec.c:1436-1437)
    t["ms_present"] = 1; t["ms_mask"][0][10] = 1
    out = oracle.spectral_tools_batch(2, c, t)
    assert np.array_equal(out[0, 1, off[10]:off[11]], np.float32(-0.5) * c[0, 0, off[10]:off[11]])


def test_tns_zero_filter_is_identity_and_allpole_inverts_fir(pkg, oracle):
    t = _long_frame(pkg)
    tns = t["ch"][0][0]["tns"]
    tns["present"] = 1; tns["n_filt"][0] = 1; tns["length"][0][0] = 49; tns["order"][0][0] = 6
    rng = np.random.default_rng(2)
    c = rng.standard_normal((1, 2, 1024)).astype(np.float32)
    assert np.array_equal(oracle.spectral_tools_batch(2, c, t), c)        # all-zero reflection coefs
    k = np.sin(np.array([3, -2, 1, 4, -1, 2]) * np.pi / 17).astype(np.float32)
    tns["coef"][0][0][:6] = k
    out = oracle.spectral_tools_batch(2, c, t)
    # Levinson step-up in double, then the matching FIR (the encoder side) on the filtered range
    lpc = np.zeros(6)
    for i in range(6):
        r = -float(k[i]); prev = lpc.copy(); lpc[i] = r
        for j in range(i):
            lpc[j] = prev[j] + r * prev[i - 1 - j]
    end = _synth().SWB_1024_48[40]                   # min(top, tns_max_bands)
    y = out[0, 0, :end].astype(np.float64)
    x = y.copy()
    for m in range(end):
        for i in range(1, min(m, 6) + 1):
            x[m] += y[m - i] * lpc[i - 1]
    assert np.max(np.abs(x - c[0, 0, :end])) < 1e-3
    assert np.array_equal(out[0, 0, end:], c[0, 0, end:]) and np.array_equal(out[0, 1], c[0, 1])
    assert not np.array_equal(out[0, 0, :end], c[0, 0, :end])


def test_noise_bands_carry_the_scalefactor_energy(pkg, oracle):
    t = _long_frame(pkg)
    t["ch"][0][0]["band_type"][20] = 13; t["ch"][0][0]["sf"][20] = 3.0
    t["ch"][0][1]["band_type"][5] = 13;  t["ch"][0][1]["sf"][5] = 0.125
    c = np.zeros((1, 2, 1024), np.float32)
    out, rs = oracle.spectral_tools_batch(2, c, t, np.array([0x1f2e3d4c], np.int32))
    off = _synth().SWB_1024_48
    e0 = float(np.sum(out[0, 0, off[20]:off[21]].astype(np.float64) ** 2))
    e1 = float(np.sum(out[0, 1, off[5]:off[6]].astype(np.float64) ** 2))
    assert abs(e0 - 9.0) < 1e-4 and abs(e1 - 0.125 ** 2) < 1e-7
    # the generator advanced by exactly the number of noise coefficients, channel 0 first
    x = 0x1f2e3d4c
    first = None
    for k in range((off[21] - off[20]) + (off[6] - off[5])):
        x = (x * 1664525 + 1013904223) & 0xffffffff
        if k == 0:
            first = x
    assert int(rs[0]) & 0xffffffff == x
    # first noise sample is the first draw (as a signed int) times the band's scale
    sgn = first - (1 << 32) if first >= (1 << 31) else first
    assert np.sign(out[0, 0, off[20]]) == np.sign(sgn)
    # everything else untouched
    keep = np.ones((2, 1024), bool); keep[0, off[20]:off[21]] = False; keep[1, off[5]:off[6]] = False
    assert not out[0][keep].any()


def test_main_prediction_tracks_a_stationary_line(pkg, oracle):
    """A spectral line that repeats frame after frame becomes predictable: with output enabled and a
    zero residual the reconstructed value approaches the line (backward-adaptive LMS)."""
    t = _long_frame(pkg)
    for c in range(2):
        pr = t["ch"][0][c]["pred"]
        pr["pred_sfb_max"] = 40; pr["predictor_present"] = 1; pr["prediction_used"][:41] = 1
    pred = np.zeros((1, 2, pkg.MAX_PREDICTORS), pkg.PRED_STATE_DT)
    pred["var0"] = 1.0; pred["var1"] = 1.0
    pred = pred.view(np.float32).reshape(1, 2, pkg.MAX_PREDICTORS, 6)
    sf = 1.0 / (1024.0 * 32768.0)
    line = np.zeros((1, 2, 1024), np.float32); line[0, :, 10] = 2000 * sf
    # phase 1: the encoder sends the full line (prediction off) for a while -> predictors adapt
    t["ch"][0][0]["pred"]["predictor_present"] = 0; t["ch"][0][1]["pred"]["predictor_present"] = 0
    for _ in range(30):
        out, pred = oracle.spectral_tools_batch(2, line, t, None, pred)
        assert np.array_equal(out, line)                         # output_enable = 0: spectrum untouched
    assert pred[0, 0, 10, 2] > 1 and pred[0, 0, 11, 2] < 1       # var0 grows where energy is, decays elsewhere
    # phase 2: residual zero, prediction on -> the decoder output is the prediction itself
    t["ch"][0][0]["pred"]["predictor_present"] = 1
    out, pred2 = oracle.spectral_tools_batch(2, np.zeros_like(line), t, None, pred)
    assert abs(out[0, 0, 10] / line[0, 0, 10] - 1) < 0.2         # within 20 % of the stationary line
    assert out[0, 1, 10] == 0                                    # channel 1 still has prediction off
    # reset group 11 (lines 10, 40, 70 ...) returns those predictors to the initial state
    t["ch"][0][0]["pred"]["predictor_reset_group"] = 11
    _, pred3 = oracle.spectral_tools_batch(2, line, t, None, pred)
    assert tuple(pred3[0, 0, 10]) == (0, 0, 1, 1, 0, 0) and tuple(pred3[0, 0, 40]) == (0, 0, 1, 1, 0, 0)
    assert pred3[0, 0, 9, 2] < 1                                 # neighbours keep adapting
