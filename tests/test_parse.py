"""The host-side AAC parser (include/heaac_parse.h, csrc/aac_parse.c).  The reference decoder cannot run here
and its tree holds no AAC vectors, so the parser is pinned three ways: (1) the ISO tables it
reads carry a committed fingerprint; (2) access units written by an independent bit writer from known side
info (tests/aac_bitwriter.py) must come back field for field, and the dequantised spectrum must equal
sign |q|^(4/3) 2^((sf - 200) / 4) computed here in float32 the way aacdec.c:988-1245 multiplies it; (3) the
AudioSpecificConfig bytes the survey recorded from the running reference parse to what it saw."""
import hashlib
import importlib
import os

import numpy as np
import pytest

import aac_bitwriter as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(pkg, aot=2, si=3, ch=2):
    c = pkg.AacConfig()
    c.object_type, c.sampling_index, c.sample_rate, c.chan_config = aot, si, [96000, 88200, 64000, 48000, 44100,
        32000, 24000, 22050, 16000, 12000, 11025, 8000, 7350][si], ch
    return c


def _write_au(rng, si, aot, cpe, extras=True, sbr=None, quiet=False):
    """One access unit + everything the parser must report for it.  sbr = (payload bits, crc): carried in a
    fill element behind the channel element (extension type 0xd / 0xe, aacdec.c:1655-1678)."""
    bw = W.BitWriter()
    exp = dict(channels=2 if cpe else 1)
    if extras and rng.random() < 0.5:
        # data_stream_element in front: skipped (aacdec.c:602-620)
        bw.put(4, 3); bw.put(int(rng.integers(0, 16)), 4)
        align = int(rng.integers(0, 2)); cnt = int(rng.integers(0, 6))
        bw.put(align, 1); bw.put(cnt, 8)
        if align:
            bw.align()
        for _ in range(cnt):
            bw.put(int(rng.integers(0, 256)), 8)
    ch = [W.random_ics(rng, si, aot, allow_intensity=False, quiet=quiet)]
    if cpe:
        bw.put(1, 3); bw.put(0, 4)
        common = int(rng.integers(0, 2))
        bw.put(common, 1)
        exp["common_window"] = common
        if common:
            second = W.random_ics(rng, si, aot, allow_intensity=True, quiet=quiet)
            for k in ("window_sequence", "window_shape", "max_sfb", "eight", "off", "num_swb", "group_len",
                      "predictor_present"):
                second[k] = ch[0][k]
            for k in ("grouping", "reset_group", "prediction_used"):
                if k in ch[0]:
                    second[k] = ch[0][k]
            # the second channel's own draws must fit the shared window: redraw with the shared layout
            second = _redraw_like(rng, ch[0], si, aot, quiet)
            ch.append(second)
            W.put_ics_info(bw, ch[0], si, aot)
            ms_present = int(rng.integers(0, 3))
            bw.put(ms_present, 2)
            nb = len(ch[0]["group_len"]) * ch[0]["max_sfb"]
            mask = np.zeros(128, np.uint8)
            if ms_present == 1:
                mask[:nb] = rng.integers(0, 2, nb)
                for v in mask[:nb]:
                    bw.put(int(v), 1)
            elif ms_present == 2:
                mask[:nb] = 1
            exp["ms_present"], exp["ms_mask"] = ms_present, mask
        else:
            ch.append(W.random_ics(rng, si, aot, allow_intensity=True, quiet=quiet))
            exp["ms_present"], exp["ms_mask"] = 0, np.zeros(128, np.uint8)
        exp["sf"] = [W.put_ics(bw, ch[0], si, aot, common), W.put_ics(bw, ch[1], si, aot, common)]
    else:
        bw.put(0, 3); bw.put(0, 4)
        exp["sf"] = [W.put_ics(bw, ch[0], si, aot, 0)]
    exp["sbr_bit"] = -1
    if sbr is not None:
        bits, crc = sbr
        cnt = (4 + len(bits) + 7) // 8
        bw.put(6, 3)
        if cnt >= 15:
            bw.put(15, 4); bw.put(cnt - 14, 8)
        else:
            bw.put(cnt, 4)
        bw.put(0xe if crc else 0xd, 4)
        exp["sbr_bit"], exp["sbr_bytes"] = len(bw.bits), cnt
        bw.bits.extend(bits)
        bw.bits.extend([0] * (8 * cnt - 4 - len(bits)))
    elif extras and rng.random() < 0.5:
        # fill element carrying an SBR payload (type 0xd): located by this parser (random bits here)
        cnt = int(rng.integers(1, 20))
        bw.put(6, 3)
        if cnt >= 15:
            bw.put(15, 4); bw.put(cnt - 14, 8)
        else:
            bw.put(cnt, 4)
        bw.put(0xd, 4)
        exp["sbr_bit"], exp["sbr_bytes"] = len(bw.bits), cnt
        for _ in range(8 * cnt - 4):
            bw.put(int(rng.integers(0, 2)), 1)
    bw.put(7, 3)
    exp["bits"] = len(bw.bits)
    exp["ch"] = ch
    return bw.bytes(), exp


def _redraw_like(rng, first, si, aot, quiet=False):
    """A second channel for a common-window pair: same ics_info, own sections / scalefactors / spectrum."""
    for _ in range(200):
        d = W.random_ics(rng, si, aot, allow_intensity=True, quiet=quiet)
        if d["eight"] == first["eight"]:
            break
    for k in ("window_sequence", "window_shape", "max_sfb", "group_len", "predictor_present"):
        d[k] = first[k]
    for k in ("grouping", "reset_group", "prediction_used"):
        if k in first:
            d[k] = first[k]
        else:
            d.pop(k, None)
    # sections and spectrum drawn for ANOTHER max_sfb / grouping do not fit: regenerate them on the shared layout
    ng, ms, off = len(d["group_len"]), d["max_sfb"], d["off"]
    bt = np.zeros((ng, ms), int)
    for g in range(ng):
        k = 0
        while k < ms:
            ln = int(rng.integers(1, ms - k + 1))
            bt[g, k:k + ln] = int(rng.choice([0] + list(range(1, 12)) + [13, 14, 15])); k += ln
    d["band_type"] = bt
    d["sf_delta"] = rng.integers(-2, 3, (ng, ms)) if quiet else rng.integers(-6, 7, (ng, ms))
    q = {}
    for g in range(ng):
        for i in range(ms):
            b = bt[g, i]
            if 1 <= b <= 11:
                lav = W.LAV[b] if b < 11 else 15
                q[(g, i)] = rng.integers(-lav, lav + 1, (d["group_len"][g], off[i + 1] - off[i]))
    d["q"] = q
    if d["pulse"] and d["eight"]:
        d["pulse"] = None
    if d["tns"] and len(d["tns"]["n_filt"]) != (8 if d["eight"] else 1):
        d["tns"] = None
    return d


def _check_channel(out, f, c, d, exp_sf, si):
    t = out["tools"][f]["ch"][c]
    ng, ms = len(d["group_len"]), d["max_sfb"]
    assert t["ics"]["num_windows"] == (8 if d["eight"] else 1) and t["ics"]["max_sfb"] == ms
    assert t["ics"]["num_window_groups"] == ng and list(t["ics"]["group_len"][:ng]) == d["group_len"]
    assert t["ics"]["num_swb"] == d["num_swb"] and list(t["ics"]["swb_offset"][:d["num_swb"] + 1]) == list(d["off"])
    assert t["ics"]["tns_max_bands"] == (W.T["aac_tns_max_bands_128"] if d["eight"] else W.T["aac_tns_max_bands_1024"])[si]
    assert np.array_equal(t["band_type"][:ng * ms], d["band_type"].reshape(-1))
    assert np.array_equal(t["sf"][:ng * ms].view(np.uint32), exp_sf[:ng * ms].view(np.uint32)), "scalefactors"
    assert out["ics"][f, c]["window_sequence"][0] == d["window_sequence"]
    assert out["ics"][f, c]["use_kb_window"][0] == d["window_shape"]
    assert t["pred"]["predictor_present"] == d["predictor_present"]
    if d["predictor_present"]:
        assert t["pred"]["predictor_reset_group"] == d["reset_group"]
        assert list(t["pred"]["prediction_used"][:len(d["prediction_used"])]) == d["prediction_used"]
    assert t["tns"]["present"] == (1 if d["tns"] else 0)
    if d["tns"]:
        for w, nf in enumerate(d["tns"]["n_filt"]):
            assert t["tns"]["n_filt"][w] == nf
            for k, fl in enumerate(d["tns"]["filt"][w]):
                assert t["tns"]["length"][w][k] == fl["length"] and t["tns"]["order"][w][k] == fl["order"]
                if fl["order"]:
                    assert t["tns"]["direction"][w][k] == fl["direction"]
                    m = W.T["tns_map"][2 * fl["compress"] + d["tns"]["coef_res"][w]]
                    want = np.array([m[v] for v in fl["idx"]], np.float32)
                    assert np.array_equal(t["tns"]["coef"][w][k][:fl["order"]].view(np.uint32), want.view(np.uint32))
    want = W.expected_pulse(d, exp_sf, W.expected_coeffs(d, exp_sf))
    got = out["coeffs"][f, c]
    small = np.array([abs(int(v)) < 16 for v in np.ones(1)])      # (escape values: cbrtf may differ by an ulp from numpy)
    exact = got.view(np.uint32) == want.view(np.uint32)
    close = np.isclose(got, want, rtol=3e-7, atol=0)
    assert close.all(), np.argwhere(~close)[:5]
    # every line below the escape range, and so every book but 11's escapes, must be bit-exact
    esc = np.zeros(1024, bool)
    base = 0
    for g, gl in enumerate(d["group_len"]):
        for i in range(ms):
            if int(d["band_type"][g, i]) == 11:
                v = d["q"][(g, i)]
                for w in range(gl):
                    esc[base + 128 * w + d["off"][i]: base + 128 * w + d["off"][i + 1]] |= np.abs(v[w]) >= 16
        base += gl * 128
    if d["pulse"]:
        for p in d["pulse"]["pos"]:
            esc[p] = True                                  # pulses go through cbrtf / sqrtf as well
    assert exact[~esc].all(), np.argwhere(~exact & ~esc)[:5]


@pytest.mark.parametrize("cpe,aot,si", [(False, 2, 3), (True, 2, 3), (True, 1, 4), (False, 2, 6), (True, 2, 11)])
def test_written_access_units_come_back(pkg, cpe, aot, si):
    rng = np.random.default_rng(1000 + 7 * si + cpe)
    cfg = _cfg(pkg, aot, si, 2 if cpe else 1)
    n = 60
    aus, exps = zip(*[_write_au(rng, si, aot, cpe) for _ in range(n)])
    st = np.zeros(n, pkg.AAC_STREAM_DT)
    out = pkg.aac_parse_batch(cfg, st, list(aus), threads=3)
    assert out["failed"] == 0 and not out["status"].any(), out["status"]
    for f, e in enumerate(exps):
        assert out["info"][f]["channels"] == e["channels"] and out["info"][f]["bits_consumed"] == e["bits"]
        assert out["info"][f]["sbr_payload_bit"] == e["sbr_bit"]
        if e["sbr_bit"] >= 0:
            assert out["info"][f]["sbr_payload_bytes"] == e["sbr_bytes"]
        if cpe:
            assert out["tools"][f]["common_window"] == e["common_window"]
            assert out["tools"][f]["ms_present"] == e["ms_present"]
            assert np.array_equal(out["tools"][f]["ms_mask"], e["ms_mask"])
        for c, d in enumerate(e["ch"]):
            _check_channel(out, f, c, d, e["sf"][c], si)


def test_window_history_is_carried_per_stream(pkg):
    """ics.window_sequence[1] / use_kb_window[1] = the previous frame's (aacdec.c:650-653); a common-window
    pair hands channel 0's history to channel 1 but keeps channel 1's own previous shape (:1462-1464)."""
    rng = np.random.default_rng(5)
    cfg = _cfg(pkg, 2, 3, 2)
    st = np.zeros(1, pkg.AAC_STREAM_DT)
    prev = [(0, 0), (0, 0)]
    for step in range(6):
        au, e = _write_au(rng, 3, 2, True, extras=False)
        out = pkg.aac_parse_batch(cfg, st, [au], threads=1)
        assert out["failed"] == 0
        for c, d in enumerate(e["ch"]):
            ic = out["ics"][0, c]
            want_prev = prev[0] if (e["common_window"] and c == 1) else prev[c]
            assert ic["window_sequence"][1] == want_prev[0]
            assert ic["use_kb_window"][1] == prev[c][1]
        prev = [(d["window_sequence"], d["window_shape"]) for d in e["ch"]]


def test_bad_streams_are_refused_and_leave_the_stream_state_alone(pkg):
    rng = np.random.default_rng(9)
    cfg = _cfg(pkg, 2, 3, 1)
    au, e = _write_au(rng, 3, 2, False, extras=False)
    st = np.zeros(1, pkg.AAC_STREAM_DT); st["window_sequence"][0, 0] = 3
    cut = au[: max(2, e["bits"] // 16)]                    # truncated: runs off the end
    out = pkg.aac_parse_batch(cfg, st, [cut], threads=1)
    assert out["failed"] == 1 and out["status"][0] in (-1, -2) and st["window_sequence"][0, 0] == 3
    bw = W.BitWriter(); bw.put(2, 3); bw.put(0, 4); bw.put(0, 40)          # coupling channel element
    assert pkg.aac_parse_batch(cfg, st, [bw.bytes()], threads=1)["status"][0] == -3
    bw = W.BitWriter(); bw.put(0, 3); bw.put(0, 4); bw.put(120, 8)         # SCE: global gain
    bw.put(0, 1); bw.put(0, 2); bw.put(0, 1); bw.put(3, 6); bw.put(0, 1)  # ics_info: long, max_sfb 3
    bw.put(12, 4); bw.put(3, 5)                                            # section with the reserved band type 12
    bw.put(0, 64)
    assert pkg.aac_parse_batch(cfg, st, [bw.bytes()], threads=1)["status"][0] == -1
    bw = W.BitWriter(); bw.put(7, 3)                                       # END without any channel element
    assert pkg.aac_parse_batch(cfg, st, [bw.bytes()], threads=1)["status"][0] == -1


def test_audio_specific_configs_of_the_survey(pkg):
    """the ASC bytes the surveyor fed the running reference and what it reported for them."""
    c, off = pkg.asc_parse(bytes([0x11, 0x90]))            # AAC-LC 48 kHz stereo
    assert (c.object_type, c.sampling_index, c.sample_rate, c.chan_config, c.sbr, c.ps) == (2, 3, 48000, 2, -1, 0)
    assert off == 13
    c, _ = pkg.asc_parse(bytes([0x11, 0x88]))              # AAC-LC 48 kHz mono
    assert (c.object_type, c.chan_config, c.sbr, c.ps) == (2, 1, -1, -1)         # implicit PS stays possible
    c, _ = pkg.asc_parse(bytes([0x2B, 0x11, 0x88, 0x00]))  # HE-AACv1 24 -> 48 kHz stereo
    assert (c.object_type, c.sample_rate, c.ext_sample_rate, c.chan_config, c.sbr, c.ps) == (2, 24000, 48000, 2, 1, 0)
    c, _ = pkg.asc_parse(bytes([0xEB, 0x09, 0x88, 0x00]))  # HE-AACv2 24 -> 48 kHz mono + PS
    assert (c.object_type, c.sample_rate, c.ext_sample_rate, c.chan_config, c.sbr, c.ps) == (2, 24000, 48000, 1, 1, 1)


def test_adts_header(pkg):
    """ff_aac_parse_header (aac_parser.c:29-70): a 7-byte header built field by field."""
    bw = W.BitWriter()
    for v, n in ((0xfff, 12), (0, 1), (0, 2), (1, 1), (1, 2), (3, 4), (0, 1), (2, 3), (0, 4), (371, 13), (0x7ff, 11), (0, 2)):
        bw.put(v, n)
    h, r = pkg.adts_parse_header(bw.bytes())
    assert r == 7 and (h.object_type, h.sampling_index, h.sample_rate, h.chan_config) == (2, 3, 48000, 2)
    assert (h.frame_length, h.samples, h.num_aac_frames, h.crc_absent) == (371, 1024, 1, 1)
    assert h.bit_rate == 371 * 8 * 48000 // 1024
    raw = bytearray(bw.bytes()); raw[0] = 0
    assert pkg.adts_parse_header(bytes(raw))[1] == -1      # no sync
    bw2 = W.BitWriter()
    for v, n in ((0xfff, 12), (0, 1), (0, 2), (1, 1), (1, 2), (13, 4), (0, 1), (2, 3), (0, 4), (371, 13), (0x7ff, 11), (0, 2)):
        bw2.put(v, n)
    assert pkg.adts_parse_header(bw2.bytes())[1] == -2     # reserved sampling frequency index
    # the parser skips an ADTS header in front of the raw data block (aacdec.c:1988-1997)
    rng = np.random.default_rng(3)
    au, e = _write_au(rng, 3, 2, True, extras=False)
    st = np.zeros(1, pkg.AAC_STREAM_DT)
    out = pkg.aac_parse_batch(_cfg(pkg), st, [bw.bytes(pad=0) + au], threads=1)
    assert out["failed"] == 0 and out["info"][0]["bits_consumed"] == 56 + e["bits"]
    # parse_adts_frame_header (aacdec.c:1935-1971): more than one raw data block per frame is unsupported, and a
    # header whose rate or object type contradicts the configuration the batch was given is refused (the reference
    # adopts the header's values; here `cfg` is read-only and a mismatch would pick the wrong band tables)
    def hdr(aot_m1, si, rdb):
        w = W.BitWriter()
        for v, n in ((0xfff, 12), (0, 1), (0, 2), (1, 1), (aot_m1, 2), (si, 4), (0, 1), (2, 3), (0, 4),
                     (7 + len(au), 13), (0x7ff, 11), (rdb, 2)):
            w.put(v, n)
        return w.bytes(pad=0)
    for h, want in ((hdr(1, 3, 0), 0), (hdr(1, 3, 1), -3), (hdr(1, 4, 0), -1), (hdr(0, 3, 0), -1)):   # OK, UNSUPPORTED, DATA, DATA
        st = np.zeros(1, pkg.AAC_STREAM_DT)
        out = pkg.aac_parse_batch(_cfg(pkg), st, [h + au], threads=1)
        assert int(out["status"][0]) == want, (want, int(out["status"][0]))


def test_iso_tables_fingerprint(pkg):
    """The Huffman codebooks and band tables the parser reads (generated header) are pinned."""
    path = os.path.join(ROOT, "ffmpeg-heaac_amd", "csrc", "aac_iso_tables.h")
    assert hashlib.sha256(open(path, "rb").read()).hexdigest() == open(
        os.path.join(ROOT, "tests", "golden", "aac_iso_tables.sha256")).read().strip()
    f = pkg.lib().heaac_aac_tables_fingerprint
    f.restype = __import__("ctypes").c_uint64
    assert f() == int(open(os.path.join(ROOT, "tests", "golden", "aac_iso_tables.fnv")).read().strip(), 16)
    # Kraft equality: every book is a complete prefix code
    first = W.T["aac_spec_first"]
    for b in range(11):
        assert sum(2.0 ** -l for l in W.T["aac_spec_bits"][first[b]:first[b + 1]]) == 1.0
    assert sum(2.0 ** -l for l in W.T["aac_sf_bits"]) == 1.0


@pytest.mark.gpu
def test_bitstream_to_pcm_on_the_gpu(pkg, oracle, dev):
    """The whole AAC-LC chain the way a host would drive it: access units -> heaac_aac_parse_batch ->
    heaac_spectral_tools_batch (noise substitution, M/S, intensity, TNS) -> heaac_lc_decode_batch, three
    frames per stream with state chained, against the oracle's tools + decode on the same parsed records."""
    import torch
    rng = np.random.default_rng(77)
    cfg = _cfg(pkg, 2, 3, 2)
    n = 48
    st = np.zeros(n, pkg.AAC_STREAM_DT)
    d_state = torch.zeros((n, 1024), device="cuda")
    ref_state = np.zeros((n, 1024), np.float32)
    d_rng = torch.full((n,), 0x1f2e3d4c, dtype=torch.int32, device="cuda")
    ref_rng = np.full(n, 0x1f2e3d4c, np.int32)
    for step in range(3):
        aus = [_write_au(rng, 3, 2, True)[0] for _ in range(n)]
        out = pkg.aac_parse_batch(cfg, st, aus)
        assert out["failed"] == 0
        ref_c, ref_rng = oracle.spectral_tools_batch(2, out["coeffs"], out["tools"], rng=ref_rng)
        ref_pcm, ref_state = oracle.lc_decode_batch(2, ref_c, out["ics"], ref_state, oracle.PCM_S16)
        d_c = torch.from_numpy(out["coeffs"]).cuda()
        dev.spectral_tools(2, d_c, pkg.to_device(out["tools"]), rng=d_rng)
        pcm, d_state = dev.lc_decode(2, d_c, pkg.to_device(out["ics"]), d_state, pcm_format=pkg.PCM_S16)
        assert np.array_equal(d_c.cpu().numpy().view(np.uint32), ref_c.view(np.uint32)), step
        assert np.array_equal(pcm.cpu().numpy(), ref_pcm), step
        assert np.array_equal(d_state.cpu().numpy().view(np.uint32), ref_state.view(np.uint32)), step
        assert np.array_equal(d_rng.cpu().numpy(), ref_rng), step
