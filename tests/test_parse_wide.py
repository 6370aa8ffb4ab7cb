"""heaac_aac_parse_frame_ex (csrc/aac_parse.c): access units with coupling channel elements (decode_cce,
aacdec.c:1503-1570), program config elements (decode_pce :303-357, read past) and packed mono spectra, written by
the test bit writer and compared field by field; the gain lists a coupling element lands on the output element are
stated here from the reference's index walk (apply_channel_coupling :1870-1898), in double / float as decode_cce
forms them."""
import json
import os

import numpy as np
import pytest

import aac_bitwriter as W
import test_parse as TP


# Every gain decode_cce can form, minted by tests/golden/make_cce_gains.c from the reference's expressions with ITS
# declared types (`float scale`, aacdec.c:1508: the base is rounded to float before pow) -- not by this file, not by
# the parser: 348 of the 484 values differ from what a double base gives.
with open(os.path.join(os.path.dirname(__file__), "golden", "cce_gains.json")) as _f:
    _G = json.load(_f)
GAIN = {False: np.array(_G["positive"], np.uint32).view(np.float32),
        True: np.array(_G["negative"], np.uint32).view(np.float32)}


def cce_gain(scale_idx, step, negative=False):
    return GAIN[bool(negative)][scale_idx, step + 128]


def write_cce(bw, rng, si, aot, tag, targets, point, quiet=True, common_gains=None, scale_idx=None, sign=None):
    """coupling_channel_element(); returns (ics dict, expected sf, gain lists [num_gain][120] float32, num_gain).
    targets: [(is_cpe, tag, ch_select)], point: 0 BEFORE_TNS, 1 BETWEEN_TNS_AND_IMDCT, 3 AFTER_IMDCT."""
    bw.put(2, 3); bw.put(tag, 4)
    bw.put(1 if point == 3 else 0, 1)                      # ind_sw_cce_flag
    bw.put(len(targets) - 1, 3)
    num_gain = 0
    for is_cpe, t, sel in targets:
        bw.put(is_cpe, 1); bw.put(t, 4)
        num_gain += 1
        if is_cpe:
            bw.put(sel, 2)
            num_gain += sel == 3
    bw.put(int(rng.integers(0, 2)) if point == 3 else point, 1)    # cc_domain (ignored when independently switched)
    sign = int(rng.integers(0, 2)) if sign is None else sign
    scale_idx = int(rng.integers(0, 4)) if scale_idx is None else scale_idx
    bw.put(sign, 1); bw.put(scale_idx, 2)
    d = W.random_ics(rng, si, aot, allow_intensity=False, quiet=quiet)
    exp_sf = W.put_ics(bw, d, si, aot, 0)
    ng, ms = len(d["group_len"]), d["max_sfb"]
    lists = np.zeros((num_gain, 120), np.float32)
    for c in range(num_gain):
        cge, gain, cache = 1, 0, np.float32(1.0)
        if c:
            if point != 3:
                cge = int(rng.integers(0, 2))
                bw.put(cge, 1)
            if cge:
                gain = int(rng.integers(-12, 13)) if common_gains is None else common_gains[c]
                W.put_sf(bw, gain)
            cache = cce_gain(scale_idx, gain)
        if point == 3:
            lists[c, 0] = cache
            continue
        idx = 0
        for g in range(ng):
            for sfb in range(ms):
                if int(d["band_type"][g, sfb]) != 0:
                    if not cge:
                        t = int(rng.integers(-3, 4))
                        W.put_sf(bw, t)
                        if t:
                            gain += t
                            # gain_element_sign: the low bit of the accumulated step is the sign, the rest (arithmetic
                            # shift) the step (:1552-1556)
                            cache = cce_gain(scale_idx, gain >> 1, gain & 1) if sign else cce_gain(scale_idx, gain)
                    lists[c, idx] = cache
                idx += 1
    return d, exp_sf, lists, num_gain


def expected_links(targets, target_type_is_cpe, target_tag, lists):
    """apply_channel_coupling's index walk for the one output element: [(target_ch, gain list)]"""
    links, index = [], 0
    for is_cpe, t, sel in targets:
        sel = sel if is_cpe else 2
        if bool(is_cpe) == bool(target_type_is_cpe) and t == target_tag:
            if sel != 1:
                links.append((0, lists[index]))
                if sel != 0:
                    index += 1
            if sel != 2:
                links.append((1, lists[index])); index += 1
        else:
            index += 1 + (sel == 3)
    return links


def write_pce(bw, rng):
    bw.put(5, 3); bw.put(int(rng.integers(0, 16)), 4)
    bw.put(int(rng.integers(0, 4)), 2); bw.put(int(rng.integers(0, 12)), 4)
    counts = [int(rng.integers(0, 3)), int(rng.integers(0, 2)), int(rng.integers(0, 2))]
    lfe, assoc, cc = int(rng.integers(0, 2)), int(rng.integers(0, 3)), int(rng.integers(0, 3))
    for c in counts:
        bw.put(c, 4)
    bw.put(lfe, 2); bw.put(assoc, 3); bw.put(cc, 4)
    for _ in range(2):
        f = int(rng.integers(0, 2)); bw.put(f, 1)
        if f:
            bw.put(int(rng.integers(0, 16)), 4)
    f = int(rng.integers(0, 2)); bw.put(f, 1)
    if f:
        bw.put(int(rng.integers(0, 8)), 3)
    for c in counts:
        for _ in range(c):
            bw.put(int(rng.integers(0, 2)), 1); bw.put(int(rng.integers(0, 16)), 4)
    for _ in range(lfe + assoc):
        bw.put(int(rng.integers(0, 16)), 4)
    for _ in range(cc):
        bw.put(int(rng.integers(0, 2)), 1); bw.put(int(rng.integers(0, 16)), 4)
    bw.align()
    n = int(rng.integers(0, 9))
    bw.put(n, 8)
    for _ in range(n):
        bw.put(int(rng.integers(0, 256)), 8)


def write_target(bw, rng, si, aot, cpe, quiet=True):
    """The output element (tag 0) as test_parse writes it, without extras; returns the channel dicts + expectations."""
    ch = [W.random_ics(rng, si, aot, allow_intensity=False, quiet=quiet)]
    exp = {}
    if cpe:
        bw.put(1, 3); bw.put(0, 4); bw.put(0, 1)             # separate windows
        ch.append(W.random_ics(rng, si, aot, allow_intensity=True, quiet=quiet))
        exp["sf"] = [W.put_ics(bw, ch[0], si, aot, 0), W.put_ics(bw, ch[1], si, aot, 0)]
    else:
        bw.put(0, 3); bw.put(0, 4)
        exp["sf"] = [W.put_ics(bw, ch[0], si, aot, 0)]
    return ch, exp


def build_au(rng, si, aot, cpe, cces, pce=False):
    """cces: [(tag, targets, point, behind_target)].  Returns (bytes, expectations)."""
    bw = W.BitWriter()
    out = dict(cce={})
    if pce:
        write_pce(bw, rng)
    for tag, targets, point, behind in cces:
        if not behind:
            out["cce"][tag] = write_cce(bw, rng, si, aot, tag, targets, point) + (targets, point, behind)
    out["order"] = [t for t, _, _, b in cces if not b] + [t for t, _, _, b in cces if b]     # bitstream order
    out["ch"], out["exp"] = write_target(bw, rng, si, aot, cpe)
    if pce and rng.random() < 0.5:
        write_pce(bw, rng)
    for tag, targets, point, behind in cces:
        if behind:
            out["cce"][tag] = write_cce(bw, rng, si, aot, tag, targets, point) + (targets, point, behind)
    bw.put(7, 3)
    return bw.bytes(), out


def check(pkg, r, got, exp, cpe, si):
    assert r == 0, r
    assert int(got["info"][0]["channels"]) == (2 if cpe else 1) and int(got["info"][0]["n_cce"]) == len(exp["cce"])
    wrapped = dict(tools=got["tools"], ics=got["ics"][None], coeffs=got["coeffs"][None] if got["coeffs"].shape[0] == 2
                   else np.concatenate([got["coeffs"], np.zeros((1, 1024), np.float32)])[None])
    for c, d in enumerate(exp["ch"]):
        TP._check_channel(wrapped, 0, c, d, exp["exp"]["sf"][c], si)
    for slot, tag in enumerate(sorted(exp["cce"])):
        d, exp_sf, lists, num_gain, targets, point, behind = exp["cce"][tag]
        rec = got["cce"][slot]
        assert (rec["present"], rec["elem_id"], rec["coupling_point"], rec["behind_target"]) == (1, tag, point, int(behind))
        assert int(rec["seq"]) == exp["order"].index(tag)
        cw = dict(tools=got["cce_tools"][slot:slot + 1], ics=got["cce_ics"][slot:slot + 1][None].repeat(2, 1),
                  coeffs=np.stack([got["cce_coeffs"][slot], np.zeros(1024, np.float32)])[None])
        TP._check_channel(cw, 0, 0, d, exp_sf, si)
        ng, ms = len(d["group_len"]), d["max_sfb"]
        assert np.array_equal(rec["band_type"][:ng * ms], d["band_type"].reshape(-1))
        assert rec["ics"].tobytes() == got["cce_tools"][slot]["ch"][0]["ics"].tobytes()
        want = expected_links(targets, cpe, 0, lists)
        assert int(rec["n_links"]) == len(want), (rec["n_links"], len(want))
        for l, (tch, gl) in enumerate(want):
            assert int(rec["link"][l]["target_ch"]) == tch
            assert np.array_equal(rec["link"][l]["gain"].view(np.uint32), gl.view(np.uint32)), (tag, l)
    for slot in range(len(exp["cce"]), pkg.MAX_CCE):
        assert got["cce"][slot]["present"] == 0


@pytest.mark.parametrize("cpe", [False, True])
def test_coupling_elements_come_back_with_their_gain_lists(pkg, cpe):
    rng = np.random.default_rng(4100 + cpe)
    si, aot = 3, 2
    cfg = TP._cfg(pkg, aot, si, 2 if cpe else 1)
    seen_points, seen_links = set(), set()
    for trial in range(120):
        ncce = int(rng.integers(1, 3))
        tags = sorted(int(x) for x in rng.choice(16, ncce, replace=False))
        if rng.random() < 0.5:
            tags = tags[::-1]                                   # slots come out in ascending tag order regardless
        cces = []
        for tag in tags:
            targets = []
            for _ in range(int(rng.integers(1, 4))):
                is_cpe = int(rng.integers(0, 2))
                targets.append((is_cpe, int(rng.choice([0, 0, 1, 5])), int(rng.integers(0, 4)) if is_cpe else 2))
            point = int(rng.choice([0, 1, 3]))
            cces.append((tag, targets, point, bool(rng.integers(0, 2))))
        au, exp = build_au(rng, si, aot, cpe, cces, pce=rng.random() < 0.3)
        st = np.zeros(1, pkg.AAC_STREAM_DT)
        r, got = pkg.aac_parse_frame_ex(cfg, st, au)
        nl = [len(expected_links(t, cpe, 0, np.zeros((16, 120), np.float32))) for _, t, _, _ in cces]
        if max(nl) > pkg.MAX_CCE_LINKS:
            assert r == -3                                      # HEAAC_PARSE_ERR_UNSUPPORTED
            continue
        check(pkg, r, got, exp, cpe, si)
        seen_points |= {p for _, _, p, _ in cces}
        seen_links |= set(nl)
        # the narrow entry refuses the same unit, as before
        r2 = pkg.aac_parse_frame_ex(cfg, np.zeros(1, pkg.AAC_STREAM_DT), au, with_cce=False)[0]
        assert r2 == -3
    assert seen_points == {0, 1, 3} and {0, 1, 2} <= seen_links


def test_every_coupling_gain_is_the_reference_float(pkg):
    """All 4 x 121 common gains through an independently switched coupling element on a pair with a gain list per
    channel (the second list carries the common gain), against the table minted from the reference's typed
    expressions; a double base would fail 348 of them."""
    rng = np.random.default_rng(9)
    si, aot = 3, 2
    cfg = TP._cfg(pkg, aot, si, 2)
    wrong_if_double = 0
    for scale_idx in range(4):
        for gain in range(-60, 61):
            bw = W.BitWriter()
            write_cce(bw, rng, si, aot, 1, [(1, 0, 3)], 3, common_gains=[0, gain], scale_idx=scale_idx)
            write_target(bw, rng, si, aot, True)
            bw.put(7, 3)
            r, got = pkg.aac_parse_frame_ex(cfg, np.zeros(1, pkg.AAC_STREAM_DT), bw.bytes())
            assert r == 0
            rec = got["cce"][0]
            assert int(rec["n_links"]) == 2
            assert rec["link"][0]["gain"][0] == np.float32(1.0)
            have = rec["link"][1]["gain"][:1].view(np.uint32)[0]
            assert have == _G["positive"][scale_idx][gain + 128], (scale_idx, gain)
            wrong_if_double += np.float32((2.0 ** (2.0 ** (scale_idx - 3))) ** -gain).view(np.uint32) != have
    assert wrong_if_double == 348
    # spot values read off the reference's expression by hand: 2^(1/8) as a float is 1.09050775..., and
    # pow(that, 56) = 128.00009, not 2^7
    assert cce_gain(0, -56) == np.float32(128.000092) and cce_gain(3, -7) == np.float32(128.0)


def test_program_config_elements_are_read_past_and_mono_spectra_pack(pkg):
    rng = np.random.default_rng(77)
    si, aot = 4, 2
    cfg = TP._cfg(pkg, aot, si, 1)
    for trial in range(60):
        au, exp = build_au(rng, si, aot, False, [], pce=True)
        st = np.zeros(1, pkg.AAC_STREAM_DT)
        r, got = pkg.aac_parse_frame_ex(cfg, st, au, coeff_channels=1)
        assert got["coeffs"].shape == (1, 1024)
        check(pkg, r, got, exp, False, si)
        # the same unit through the two-channel layout gives the same spectrum in channel 0
        r2, got2 = pkg.aac_parse_frame_ex(cfg, np.zeros(1, pkg.AAC_STREAM_DT), au, coeff_channels=2)
        assert r2 == 0 and np.array_equal(got2["coeffs"][0].view(np.uint32), got["coeffs"][0].view(np.uint32))
    # a pair cannot be packed into one channel; a second output element is outside the slice
    au, _ = build_au(rng, si, aot, True, [])
    assert pkg.aac_parse_frame_ex(TP._cfg(pkg, aot, si, 2), np.zeros(1, pkg.AAC_STREAM_DT), au, coeff_channels=1)[0] == -4
    # three coupling elements (and up to one per instance tag): taken since round 4
    au, exp3 = build_au(rng, si, aot, False, [(1, [(0, 0, 2)], 0, False), (2, [(0, 0, 2)], 1, False), (3, [(0, 0, 2)], 1, True)])
    r3, got3 = pkg.aac_parse_frame_ex(cfg, np.zeros(1, pkg.AAC_STREAM_DT), au)
    check(pkg, r3, got3, exp3, False, si)
    bw = W.BitWriter()
    write_target(bw, rng, si, aot, False); write_target(bw, rng, si, aot, False); bw.put(7, 3)
    assert pkg.aac_parse_frame_ex(cfg, np.zeros(1, pkg.AAC_STREAM_DT), bw.bytes())[0] == -3


def test_two_coupling_channels_keep_their_histories_whatever_order_they_arrive_in(pkg):
    """The history is the element's (by instance tag, che[TYPE_CCE][tag] in the reference), not the slot's: tags 5 and
    2 arriving in descending order, in ascending order, or alone."""
    rng = np.random.default_rng(50)
    si, aot = 3, 2
    cfg = TP._cfg(pkg, aot, si, 1)
    st = np.zeros(1, pkg.AAC_STREAM_DT)
    prev = {}
    for t in range(24):
        tags = [[5, 2], [2, 5], [5], [2]][int(rng.integers(0, 4))]
        au, exp = build_au(rng, si, aot, False, [(tag, [(0, 0, 2)], 3, bool(rng.integers(0, 2))) for tag in tags])
        r, got = pkg.aac_parse_frame_ex(cfg, st, au)
        assert r == 0
        for slot, tag in enumerate(sorted(tags)):
            d = exp["cce"][tag][0]
            assert int(got["cce"][slot]["elem_id"]) == tag
            assert int(got["cce_ics"][slot]["window_sequence"][0]) == d["window_sequence"]
            if tag in prev:
                assert int(got["cce_ics"][slot]["window_sequence"][1]) == prev[tag]["window_sequence"], (t, tag)
                assert int(got["cce_ics"][slot]["use_kb_window"][1]) == prev[tag]["window_shape"], (t, tag)
            prev[tag] = d


def test_coupling_channel_keeps_its_own_window_history(pkg):
    """window_sequence[1] / use_kb_window[1] of a coupling channel are its own previous frame's."""
    rng = np.random.default_rng(5)
    si, aot = 3, 2
    cfg = TP._cfg(pkg, aot, si, 1)
    st = np.zeros(1, pkg.AAC_STREAM_DT)
    prev = None
    for t in range(12):
        au, exp = build_au(rng, si, aot, False, [(4, [(0, 0, 2)], 3, bool(t & 1))])
        r, got = pkg.aac_parse_frame_ex(cfg, st, au)
        assert r == 0
        d = exp["cce"][4][0]
        assert int(got["cce_ics"][0]["window_sequence"][0]) == d["window_sequence"]
        if prev is not None:
            assert int(got["cce_ics"][0]["window_sequence"][1]) == prev["window_sequence"]
            assert int(got["cce_ics"][0]["use_kb_window"][1]) == prev["window_shape"]
        prev = d
