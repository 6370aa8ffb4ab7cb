"""Coupling channel elements on the GPU: access units written by the test bit writer -> the wide
host parser -> heaac_spectral_tools_batch_ex (PRE / POST halves, dependent coupling around TNS) -> IMDCT ->
independent coupling, against the oracle's restatement of the same steps (oracle/or_tools.c:
apply_dependent_coupling, apply_channel_coupling) on the same parsed records; and the codec surface end to end."""
import ctypes as C

import numpy as np
import pytest

import test_parse as TP
import test_parse_wide as TW

pytestmark = pytest.mark.gpu


def _units(pkg, rng, si, aot, cpe, n, behind, points, two=False):
    """n access units with the same element ORDER (a batch shares it), parsed: returns stacked arrays."""
    cfg = TP._cfg(pkg, aot, si, 2 if cpe else 1)
    recs = []
    while len(recs) < n:
        cces = []
        tags = [3, 9] if two else [int(rng.integers(0, 16))]
        for tag in tags:
            targets = [(1 if cpe else 0, 0, int(rng.integers(0, 4)) if cpe else 2)]
            if rng.random() < 0.5:
                targets.insert(int(rng.integers(0, 2)), (int(rng.integers(0, 2)), 7, 3))       # a target that is not ours
            cces.append((tag, targets, int(rng.choice(points)), behind))
        au, _ = TW.build_au(rng, si, aot, cpe, cces)
        r, got = pkg.aac_parse_frame_ex(cfg, np.zeros(1, pkg.AAC_STREAM_DT), au)
        if r == 0:
            recs.append((au, got))
    stack = lambda k: np.stack([g[k] for _, g in recs])
    return [a for a, _ in recs], dict(coeffs=stack("coeffs"), tools=np.concatenate([g["tools"] for _, g in recs]),
                                      ics=stack("ics"), cce=stack("cce"), cce_coeffs=stack("cce_coeffs"),
                                      cce_tools=stack("cce_tools"), cce_ics=stack("cce_ics"))


@pytest.mark.parametrize("cpe,behind,two", [(False, False, False), (True, False, False), (True, True, False), (False, True, True)])
def test_dependent_coupling_around_tns(pkg, oracle, dev, cpe, behind, two):
    import torch
    rng = np.random.default_rng(900 + 4 * cpe + 2 * behind + two)
    si, aot, n = 3, 2, 24
    ch = 2 if cpe else 1
    _, u = _units(pkg, rng, si, aot, cpe, n, behind, [0, 1], two)
    ncce = pkg.MAX_CCE
    rng0 = np.full(n, 0x1f2e3d4c, np.int32)
    coeffs = np.ascontiguousarray(u["coeffs"][:, :ch])

    # --- oracle, in bitstream order ---
    def o_cce(r):
        out = np.zeros((n, ncce, 1024), np.float32)
        for s in range(ncce):
            c, r, _ = oracle.spectral_tools_batch_ex(1, oracle.TOOLS_ALL, u["cce_coeffs"][:, s:s + 1],
                                                     np.ascontiguousarray(u["cce_tools"][:, s]), rng=r)
            out[:, s] = c[:, 0]
        return out, r
    if not behind:
        ref_cce, r1 = o_cce(rng0)
        ref, r2, _ = oracle.spectral_tools_batch_ex(ch, oracle.TOOLS_ALL, coeffs, u["tools"], rng=r1, cce=u["cce"],
                                                    cce_coeffs=ref_cce)
    else:
        pre, r1, _ = oracle.spectral_tools_batch_ex(ch, oracle.TOOLS_PRE, coeffs, u["tools"], rng=rng0)
        ref_cce, r2 = o_cce(r1)
        ref, _, _ = oracle.spectral_tools_batch_ex(ch, oracle.TOOLS_POST, pre, u["tools"], cce=u["cce"], cce_coeffs=ref_cce)
    # coupling really happened: the result differs from the tools without it
    plain = oracle.spectral_tools_batch(ch, coeffs, u["tools"], rng=rng0 if behind else r1)[0]
    assert not np.array_equal(plain, ref)

    # --- GPU, the same calls ---
    d_c = torch.from_numpy(coeffs).cuda()
    d_t = pkg.to_device(u["tools"])
    d_rng = torch.from_numpy(rng0.copy()).cuda()
    d_cce = pkg.to_device(u["cce"])
    d_cc = torch.from_numpy(u["cce_coeffs"].copy()).cuda()

    def g_cce():
        for s in range(ncce):
            slot = d_cc[:, s].contiguous()
            dev.spectral_tools_ex(1, pkg.TOOLS_ALL, slot, pkg.to_device(np.ascontiguousarray(u["cce_tools"][:, s])), rng=d_rng)
            d_cc[:, s] = slot
    if not behind:
        g_cce()
        dev.spectral_tools_ex(ch, pkg.TOOLS_ALL, d_c, d_t, rng=d_rng, cce=d_cce, cce_coeffs=d_cc)
    else:
        dev.spectral_tools_ex(ch, pkg.TOOLS_PRE, d_c, d_t, rng=d_rng)
        g_cce()
        dev.spectral_tools_ex(ch, pkg.TOOLS_POST, d_c, d_t, cce=d_cce, cce_coeffs=d_cc)
    torch.cuda.synchronize()
    assert np.array_equal(d_cc.cpu().numpy().view(np.uint32), ref_cce.view(np.uint32))
    assert np.array_equal(d_c.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    assert np.array_equal(d_rng.cpu().numpy(), r2)


SCE, CPE, CCE, LFE = 0, 1, 2, 3
STREAMS = {
    # name: (object type, output elements in bitstream order, coupling element tags, coupling points drawn from)
    "pair_dependent": (2, [(CPE, 0)], [5], [0, 1]),
    "mono_independent": (2, [(SCE, 0)], [5], [3]),
    "five_one": (2, [(SCE, 0), (CPE, 0), (CPE, 1), (LFE, 0)], [3, 9], [0, 1, 3]),
    "main_three": (1, [(SCE, 2), (CPE, 4)], [7], [0, 1, 3]),
    # dependent coupling in an HE-AAC stream: in the spectrum, before the IMDCT and SBR of the targets
    "he_three_dependent": (2, [(SCE, 0), (CPE, 0), (LFE, 1)], [4, 11], [0, 1]),
    # ... and independent coupling there: the coupling channel goes through SBR with its own payload, and couples
    # over 2048 samples
    "he_three_all": (2, [(SCE, 0), (CPE, 0), (LFE, 1)], [4, 11], [0, 1, 3]),
}


@pytest.mark.parametrize("name", sorted(STREAMS))
def test_codec_decodes_access_units_with_coupling_elements(pkg, oracle, dev, name):
    """heaac_codec_decode on AAC-LC / Main streams whose program config element names coupling elements (the only
    streams whose coupling elements aac_decode_frame knows, che_configure aacdec.c:198-212): dependent coupling around
    every target's TNS and independent coupling behind its IMDCT, onto one and onto several output elements, the
    coupling elements anywhere between them; state chained over six frames.  int16 PCM against tests/coupled_ref.py
    (the oracle's tools, coupling, IMDCTs and interleave on the separately parsed records)."""
    import coupled_ref as R
    from test_shim_gpu import HeaacCodecContext, HeaacPacket
    lib = pkg.lib()
    aot, elems, cc_tags, points = STREAMS[name]
    rng = np.random.default_rng(sum(map(ord, name)))
    he = name.startswith("he_")
    si = 6 if he else 3
    length = 2048 if he else 1024
    asc = R.asc(aot, si, elems, cc_tags, rng, he=he)
    uw = R.UnitWriter(pkg, rng, si, aot, elems, cc_tags, points, he)
    ctx = HeaacCodecContext(cfg=-1, extradata=asc, extradata_size=len(asc))
    codec = C.c_void_p.in_dll(lib, "heaac_aac_decoder")
    assert lib.heaac_codec_open(C.byref(ctx), C.c_void_p(C.addressof(codec))) == 0
    r, m4, layout = pkg.asc_layout(asc)
    assert r == 0
    assert m4.sbr == (1 if he else -1) and m4.ps == 0 if he else True
    chk = R.Checker(pkg, oracle, m4, layout, aot, he=he)
    assert ctx.channels == chk.nch
    out = (C.c_int16 * (192000 // 2))()
    loud = 0

    def unit(tags, check=True, pts=None):
        return uw.unit(tags, chk.parses if check else None, pts)

    def decode(au):
        b = C.create_string_buffer(au, len(au))
        pkt = HeaacPacket(C.cast(b, C.c_void_p), len(au))
        size = C.c_int(192000)
        return lib.heaac_codec_decode(C.byref(ctx), out, C.byref(size), C.byref(pkt)), size.value

    for t in range(6):
        au = unit(cc_tags, pts=[3] if 3 in points and t in (1, 4) else points)     # every point is met for certain
        used, size = decode(au)
        assert used == len(au) and size == length * chk.nch * 2, (t, used)
        got = np.frombuffer(out, np.int16, length * chk.nch).reshape(length, chk.nch).copy()
        want, _ = chk.frame(au)
        assert np.array_equal(got, want), ("frame %d" % t, np.argwhere(got != want)[:4])
        loud = max(loud, int(np.abs(got.astype(int)).max()))
    assert loud > 50 and (chk.dependent >= 2 or points == [3]) and (chk.independent >= 2 or 3 not in points)
    if uw.cce_writers:
        assert chk.sbr_coupled >= 2                        # coupling channels really went through SBR
    # a coupling element the stream has carried so far is left out: refused, as an output element would be (the
    # reference couples whatever the element's buffers still hold)
    assert decode(unit([]))[0] < 0
    # a coupling element the program config element did not name: "channel element 2.%d is not allocated"
    assert decode(unit(cc_tags[:-1] + [13], check=False))[0] < 0
    assert lib.heaac_codec_close(C.byref(ctx)) == 0


def test_ltp_profile_stream_keeps_its_coupling_elements_but_couples_nothing_in_the_spectrum(pkg, oracle, dev):
    """A self-configuring ADTS stream (channel configuration 0, the program config element ahead of the first frame's
    channel elements) whose header says LTP profile: parse_adts_frame_header takes the object type as it comes
    (aacdec.c:1959), apply_dependent_coupling refuses to couple for it (:1822-1826), independent coupling goes on."""
    import coupled_ref as R
    import aac_bitwriter as W
    import test_parse_layout as TL
    from test_shim_gpu import HeaacCodecContext, HeaacPacket, _adts
    lib = pkg.lib()
    rng = np.random.default_rng(404)
    si, elems, cc_tags, points = 3, [(SCE, 1), (CPE, 3)], [2, 8], [0, 1, 3]
    ctx = HeaacCodecContext(cfg=-1, extradata=None, extradata_size=0)
    codec = C.c_void_p.in_dll(lib, "heaac_aac_decoder")
    assert lib.heaac_codec_open(C.byref(ctx), C.c_void_p(C.addressof(codec))) == 0
    pce = W.BitWriter()
    pce.put(5, 3); pce.put(0, 4)
    a = R.pce_args(elems, cc_tags)
    TL.write_pce_body(pce, np.random.default_rng(1), *a[:4], cc=a[4])
    r, layout, _ = pkg.aac_layout_from_pce(pce.bytes(), 7)
    assert r == 0
    m4 = TP._cfg(pkg, 4, si, 0)                            # object type 4: what the checker's parser must be told too
    chk = R.Checker(pkg, oracle, m4, layout, 4)
    uw = R.UnitWriter(pkg, rng, si, 2, elems, cc_tags, points, False)
    out = (C.c_int16 * (192000 // 2))()
    for t in range(6):
        # the units are written as AAC-LC ones (no predictor bit); the first carries the program config element
        while True:
            body = uw.unit(cc_tags, pts=[3] if t == 2 else None)
            if t == 0:
                # splice: program config element, byte aligned by its own rule, then the unit's bits
                bw = W.BitWriter()
                bw.bits.extend(pce.bits)
                bw.bits.extend((b >> (7 - i)) & 1 for b in body for i in range(8))
                body = bw.bytes()
            pkt_bytes = _adts(body, 4, si, 0)
            if chk.parses(pkt_bytes):
                break
        b = C.create_string_buffer(pkt_bytes, len(pkt_bytes))
        pkt = HeaacPacket(C.cast(b, C.c_void_p), len(pkt_bytes))
        size = C.c_int(192000)
        used = lib.heaac_codec_decode(C.byref(ctx), out, C.byref(size), C.byref(pkt))
        assert used == len(pkt_bytes) and size.value == 1024 * chk.nch * 2, (t, used)
        got = np.frombuffer(out, np.int16, 1024 * chk.nch).reshape(1024, chk.nch).copy()
        want, _ = chk.frame(pkt_bytes)
        assert np.array_equal(got, want), ("frame %d" % t, np.argwhere(got != want)[:4])
    assert chk.ltp_skipped >= 3 and chk.independent >= 1 and chk.dependent == 0
    assert lib.heaac_codec_close(C.byref(ctx)) == 0


@pytest.mark.parametrize("cpe", [False, True])
def test_codec_refuses_a_coupling_element_in_a_channel_configuration_stream(pkg, dev, cpe):
    """Channel configurations 1..7 allocate no coupling element (set_default_channel_config aacdec.c:359-398, get_che
    :132-177): aac_decode_frame fails such an access unit ("channel element 2.%d is not allocated", :2006-2010), and
    goes on with the next."""
    from test_shim_gpu import HeaacCodecContext, HeaacPacket
    lib = pkg.lib()
    rng = np.random.default_rng(77 + cpe)
    asc = bytes([0x11, 0x90]) if cpe else bytes([0x11, 0x88])
    ctx = HeaacCodecContext(cfg=-1, extradata=asc, extradata_size=2)
    codec = C.c_void_p.in_dll(lib, "heaac_aac_decoder")
    assert lib.heaac_codec_open(C.byref(ctx), C.c_void_p(C.addressof(codec))) == 0
    out = (C.c_int16 * (192000 // 2))()
    for behind in (False, True, None):
        cces = [] if behind is None else [(5, [(1 if cpe else 0, 0, 2)], 0, behind)]
        au, _ = TW.build_au(rng, 3, 2, cpe, cces)
        b = C.create_string_buffer(au, len(au))
        pkt = HeaacPacket(C.cast(b, C.c_void_p), len(au))
        size = C.c_int(192000)
        used = lib.heaac_codec_decode(C.byref(ctx), out, C.byref(size), C.byref(pkt))
        assert (used == len(au)) if behind is None else (used < 0)
    assert lib.heaac_codec_close(C.byref(ctx)) == 0
