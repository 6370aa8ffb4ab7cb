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


@pytest.mark.parametrize("cpe,points", [(True, [0, 1]), (False, [3]), (True, [0, 1, 3])])
def test_codec_decodes_access_units_with_coupling_elements(pkg, oracle, dev, cpe, points):
    """heaac_codec_decode on AAC-LC access units that carry a coupling element (dependent or AFTER_IMDCT), state
    chained over six frames; int16 PCM against the oracle run on the separately parsed records."""
    from test_shim_gpu import HeaacCodecContext, HeaacPacket
    lib = pkg.lib()
    rng = np.random.default_rng(300 + cpe + len(points))
    si, aot = 3, 2
    ch = 2 if cpe else 1
    asc = bytes([0x11, 0x90]) if cpe else bytes([0x11, 0x88])
    ctx = HeaacCodecContext(cfg=-1, extradata=asc, extradata_size=2)
    codec = C.c_void_p.in_dll(lib, "heaac_aac_decoder")
    assert lib.heaac_codec_open(C.byref(ctx), C.c_void_p(C.addressof(codec))) == 0
    cfg = TP._cfg(pkg, aot, si, ch)
    st = np.zeros(1, pkg.AAC_STREAM_DT)
    state = np.zeros((1, 512 * ch), np.float32)
    cstate = np.zeros((pkg.MAX_CCE, 1, 512), np.float32)
    r_state = np.full(1, 0x1f2e3d4c, np.int32)
    out = (C.c_int16 * (192000 // 2))()
    coupled = loud = 0
    for t in range(6):
        behind = bool(t & 1)
        while True:
            targets = [(1 if cpe else 0, 0, int(rng.integers(0, 4)) if cpe else 2)]
            au, _ = TW.build_au(rng, si, aot, cpe, [(5, targets, int(rng.choice(points)), behind)])
            r, g = pkg.aac_parse_frame_ex(cfg, st.copy(), au)
            if r == 0:
                break
        r, g = pkg.aac_parse_frame_ex(cfg, st, au)
        b = C.create_string_buffer(au, len(au))
        pkt = HeaacPacket(C.cast(b, C.c_void_p), len(au))
        size = C.c_int(192000)
        assert lib.heaac_codec_decode(C.byref(ctx), out, C.byref(size), C.byref(pkt)) == len(au), t
        assert size.value == 1024 * ch * 2 and ctx.channels == ch
        got = np.frombuffer(out, np.int16, 1024 * ch).reshape(1024, ch).copy()
        # the oracle: elements' tools in bitstream order, coupling, IMDCTs, independent coupling
        cce, cc = g["cce"][None], g["cce_coeffs"][None].copy()
        coeffs = np.ascontiguousarray(g["coeffs"][None, :ch])

        def o_cce(rs):
            for s in range(pkg.MAX_CCE):
                if cce[0, s]["present"]:
                    c1, rs, _ = oracle.spectral_tools_batch_ex(1, oracle.TOOLS_ALL, cc[:, s:s + 1], g["cce_tools"][s:s + 1], rng=rs)
                    cc[:, s] = c1[:, 0]
            return rs
        if not behind:
            r_state = o_cce(r_state)
            pre, r_state, _ = oracle.spectral_tools_batch_ex(ch, oracle.TOOLS_PRE, coeffs, g["tools"], rng=r_state)
        else:
            pre, r_state, _ = oracle.spectral_tools_batch_ex(ch, oracle.TOOLS_PRE, coeffs, g["tools"], rng=r_state)
            r_state = o_cce(r_state)
        post, _, _ = oracle.spectral_tools_batch_ex(ch, oracle.TOOLS_POST, pre, g["tools"], cce=cce, cce_coeffs=cc)
        ics = np.ascontiguousarray(g["ics"][None, :ch])
        f32, state = oracle.lc_decode_batch(ch, post, ics, state, oracle.PCM_F32)
        ref16 = None
        for s in range(pkg.MAX_CCE):
            rec = cce[0, s]
            if rec["present"] and rec["coupling_point"] == 3:
                ret, cstate[s] = oracle.lc_decode_batch(1, cc[:, s:s + 1], g["cce_ics"][s:s + 1][None], cstate[s], oracle.PCM_F32)
                for l in range(max(1, int(rec["n_links"]))):
                    cpl = np.zeros(1, pkg.COUPLING_DT)
                    if l < rec["n_links"]:
                        cpl["on"][0, rec["link"][l]["target_ch"]] = 1
                        cpl["gain"][0, rec["link"][l]["target_ch"]] = rec["link"][l]["gain"][0]
                    f32, ref16 = oracle.couple_after_imdct_batch(ch, f32, ret.reshape(1, 1024), cpl, s16=True)
        if ref16 is None:       # nothing couples behind the IMDCT: float_to_int16_interleave of the target alone
            ref16 = oracle.couple_after_imdct_batch(ch, f32, np.zeros((1, 1024), np.float32), np.zeros(1, pkg.COUPLING_DT), s16=True)[1]
        assert np.array_equal(got, ref16[0]), "frame %d" % t
        coupled += int(cce[0, 0]["n_links"]) > 0
        loud = max(loud, int(np.abs(got.astype(int)).max()))
    assert coupled >= 3 and loud > 50
    assert lib.heaac_codec_close(C.byref(ctx)) == 0
