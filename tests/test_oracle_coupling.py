"""The oracle's dependent coupling (oracle/or_tools.c: or_dependent_coupling / or_channel_coupling, after
aacdec.c:1813-1843, 1870-1898) against what it is by definition: with the target's TNS switched off, the POST half adds
gain[idx] * coupling spectrum over the coupling channel's non-zero bands (its own grouping and band offsets) into the
linked target channels, once per link, in slot order -- written down in numpy."""
import numpy as np

import test_parse as TP
import test_parse_wide as TW


def test_dependent_coupling_is_gain_times_spectrum_per_band(pkg, oracle):
    rng = np.random.default_rng(11)
    si, aot = 3, 2
    hits = 0
    for trial in range(40):
        cpe = bool(trial & 1)
        ch = 2 if cpe else 1
        cfg = TP._cfg(pkg, aot, si, ch)
        targets = [(1 if cpe else 0, 0, int(rng.integers(0, 4)) if cpe else 2)]
        au, _ = TW.build_au(rng, si, aot, cpe, [(2, targets, int(rng.integers(0, 2)), False),
                                                 (11, targets[::-1], int(rng.integers(0, 2)), True)])
        r, g = pkg.aac_parse_frame_ex(cfg, np.zeros(1, pkg.AAC_STREAM_DT), au)
        if r != 0:
            continue
        tools = g["tools"].copy()
        tools["ch"]["tns"]["present"] = 0
        coeffs = np.ascontiguousarray(g["coeffs"][None, :ch])
        cce, cc = g["cce"][None], g["cce_coeffs"][None]
        got, _, _ = oracle.spectral_tools_batch_ex(ch, oracle.TOOLS_POST, coeffs, tools, cce=cce, cce_coeffs=cc)
        want = coeffs.copy()
        for point in (0, 1):
            for s in range(pkg.MAX_CCE):
                rec = cce[0, s]
                if not rec["present"] or rec["coupling_point"] != point:
                    continue
                ics = rec["ics"]
                for l in range(int(rec["n_links"])):
                    tch = int(rec["link"][l]["target_ch"])
                    base, idx = 0, 0
                    for gi in range(int(ics["num_window_groups"])):
                        gl = int(ics["group_len"][gi])
                        for sfb in range(int(ics["max_sfb"])):
                            if rec["band_type"][idx] != 0:
                                gain = rec["link"][l]["gain"][idx]
                                for w in range(gl):
                                    lo, hi = base + 128 * w + int(ics["swb_offset"][sfb]), base + 128 * w + int(ics["swb_offset"][sfb + 1])
                                    want[0, tch, lo:hi] = want[0, tch, lo:hi] + np.float32(gain) * cc[0, s, lo:hi]
                                    hits += hi > lo
                            idx += 1
                        base += gl * 128
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), trial
    assert hits > 200
