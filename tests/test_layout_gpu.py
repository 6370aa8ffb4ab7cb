"""Channel layouts on the GPU: heaac_pcm_interleave_batch against ff_float_to_int16_interleave_c (dsputil.c:3989-4001,
oracle/or_core.c), and whole multi-element streams through heaac_codec_decode against the oracle decoding the same
elements one by one."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("channels,length", [(1, 1024), (2, 2048), (3, 1024), (6, 1024), (6, 2048), (8, 1024), (16, 1024)])
@pytest.mark.parametrize("sse2", [False, True])
def test_interleave_any_channel_count(pkg, oracle, dev, channels, length, sse2):
    import torch
    rng = np.random.default_rng(17 * channels + length + sse2)
    n = 5
    # planes as a layout's elements leave them: pairs in [n][2][len] buffers, single channels in [n][1][len]
    bufs, planes, host = [], [], []
    c = 0
    while c < channels:
        pair = c + 1 < channels and rng.random() < 0.6
        k = 2 if pair else 1
        if sse2:
            a = rng.standard_normal((n, k, length)).astype(np.float32) * 20000.0
            a.reshape(-1)[::97] = 70000.0; a.reshape(-1)[5::131] = -70000.0; a.reshape(-1)[3] = np.nan
        else:
            a = (385.0 + rng.standard_normal((n, k, length)) * 0.4).astype(np.float32)       # the C path's biased floats
            a.reshape(-1)[::89] = 386.5; a.reshape(-1)[7::113] = 383.0                  # beyond full scale: saturates
        t = torch.from_numpy(a).cuda()
        bufs.append(t)
        for j in range(k):
            planes.append((t, j * length, k * length))
            host.append(a[:, j])
        c += k
    got = dev.pcm_interleave(planes, length, pkg.PCM_S16_SSE2 if sse2 else pkg.PCM_S16).cpu().numpy()
    for f in range(n):
        want = oracle.float_to_int16_interleave([h[f] for h in host], sse2=sse2)
        assert np.array_equal(got[f], want), (f, np.argwhere(got[f] != want)[:4])


def test_interleave_refuses_what_it_cannot_do(pkg, dev):
    import torch
    t = torch.zeros(4096, device="cuda")
    L = pkg.lib()
    refs = (pkg._PlaneRef * 2)()
    out = torch.zeros(8192, dtype=torch.int16, device="cuda")
    refs[0].d_base = t.data_ptr(); refs[0].frame_stride = 1024
    refs[1].d_base = t.data_ptr() + 4; refs[1].frame_stride = 1024          # not 16-byte aligned
    args = lambda ch, ln, fmt: L.heaac_pcm_interleave_batch(dev._h, ch, refs, ln, fmt, C.c_void_p(out.data_ptr()), C.c_size_t(2), None)
    assert args(2, 1024, pkg.PCM_S16) == -1
    assert args(1, 1022, pkg.PCM_S16) == -1 and args(1, 1024, pkg.PCM_F32) == -1 and args(17, 1024, pkg.PCM_S16) == -1
    assert args(1, 1024, pkg.PCM_S16) == 0


SCE, CPE, CCE, LFE = 0, 1, 2, 3


def _asc(aot, si, cc, he=False, pce=None, ext_si=None):
    """AudioSpecificConfig (mpeg4audio.c:79-143): explicit SBR puts object type 5 and the extension rate in front."""
    import aac_bitwriter as W
    bw = W.BitWriter()
    if he:
        bw.put(5, 5); bw.put(si, 4); bw.put(cc, 4); bw.put(si - 3 if ext_si is None else ext_si, 4); bw.put(aot, 5)
    else:
        bw.put(aot, 5); bw.put(si, 4); bw.put(cc, 4)
    bw.put(0, 3)                                                       # GASpecificConfig: 1024 samples, no core coder, no extension
    if pce is not None:
        bw.put(0, 4)
        pce(bw)
    return bw.bytes()


MODES = {
    # name: (channel configuration, elements in bitstream order, SBR, how the stream configures itself)
    "lc_3_0": (3, [(SCE, 0), (CPE, 0)], False, "asc"),
    "lc_5_1": (6, [(SCE, 0), (CPE, 0), (CPE, 1), (LFE, 0)], False, "asc"),
    "lc_7_1_adts": (7, [(SCE, 0), (CPE, 0), (CPE, 1), (CPE, 2), (LFE, 0)], False, "adts"),
    "main_5_0": (5, [(SCE, 3), (CPE, 1), (CPE, 0)], False, "asc"),      # tags as the encoder pleases: taken by position
    "he_5_1": (6, [(SCE, 0), (CPE, 0), (CPE, 1), (LFE, 0)], True, "asc"),
    "he_5_1_implicit": (6, [(SCE, 0), (CPE, 0), (CPE, 1), (LFE, 0)], True, "asc_implicit"),
    "he_5_0_downsampled": (5, [(SCE, 0), (CPE, 0), (CPE, 1)], True, "asc_downsampled"),   # extension rate = core rate
    "lc_pce_asc": (0, [(CPE, 1), (LFE, 2), (SCE, 0), (CPE, 0)], False, "asc"),
    "lc_pce_adts": (0, [(SCE, 0), (CPE, 0), (CPE, 1), (LFE, 0)], False, "adts"),
    # a program config element + explicitly signalled SBR: ps = -1 -> 1 (aacdec.c:476-477) and every SCE of the
    # layout gets a second, Parametric Stereo output channel (che_configure :203-206)
    "he_pce_ps_3_0": (0, [(SCE, 0), (CPE, 0)], True, "asc"),
    "he_pce_ps_mono": (0, [(SCE, 0)], True, "asc"),
    "he_pce_ps_5_1_downsampled": (0, [(SCE, 0), (CPE, 0), (CPE, 1), (LFE, 0)], True, "asc_downsampled"),
    # ... and a ONE-channel program-config stream whose first access unit carries an SBR payload: implicit SBR turns
    # Parametric Stereo on with it and the output is configured again with two channels (aacdec.c:1670-1673)
    "he_pce_ps_mono_implicit": (0, [(SCE, 0)], True, "asc_implicit"),
}


@pytest.mark.parametrize("mode", sorted(MODES))
def test_codec_decodes_multichannel_streams(pkg, oracle, dev, mode):
    """aac_decode_frame for layouts with several output elements, through heaac_codec_decode on the reference's own
    AVCodecContext / AVPacket records: 3.0, 5.0, 5.1, 7.1, a program config element in the extradata and one at the
    head of the first ADTS frame, AAC-Main prediction, explicit and implicit SBR per element, downsampled SBR (1024
    samples per frame at the core rate, aacsbr.c:1719), Parametric Stereo on the single channel elements of a
    program-config layout with explicit SBR (two output channels each).  The checker parses the
    same units with the layout parser (pinned by tests/test_parse_layout.py), runs the ORACLE's spectral tools through
    the elements in bitstream order (one noise generator), the oracle's decode per element on its own state, and
    ff_float_to_int16_interleave_c over the planes in layout order."""
    import copy
    import sbr_bitwriter as SW
    import test_parse as TP
    import test_parse_layout as TL
    from test_shim_gpu import HeaacCodecContext, HeaacPacket, _adts
    lib = pkg.lib()
    cc, elems, he, how = MODES[mode]
    rng = np.random.default_rng(sum(map(ord, mode)))
    aot = 1 if mode.startswith("main") else 2
    si = 6 if he else 3
    pce_elems = None
    if cc == 0:
        front = [(int(t == CPE), g) for t, g in elems if t in (SCE, CPE) and not (t == CPE and g == 1)]
        back = [(1, 1)] if (CPE, 1) in elems else []
        pce_elems = (front, [], back, [g for t, g in elems if t == LFE])
    pce = (lambda bw: TL.write_pce_body(bw, rng, *pce_elems)) if cc == 0 else None
    down = how == "asc_downsampled"
    asc = None if how == "adts" else _asc(aot, si, cc, he=he and how in ("asc", "asc_downsampled"), pce=pce, ext_si=si if down else None)
    ctx = HeaacCodecContext(cfg=-1, extradata=asc, extradata_size=len(asc) if asc else 0)
    codec = C.c_void_p.in_dll(lib, "heaac_aac_decoder")
    assert lib.heaac_codec_open(C.byref(ctx), C.c_void_p(C.addressof(codec))) == 0
    # the checker's own configuration, layout and per-element streams
    m4 = TP._cfg(pkg, aot, si, cc)
    if cc:
        r, layout = pkg.aac_layout_default(cc)
    else:
        bw = __import__("aac_bitwriter").BitWriter()
        TL.write_pce_body(bw, np.random.default_rng(1), *pce_elems)
        r, layout, _ = pkg.aac_layout_from_pce(bw.bytes(), 0)
    assert r == 0
    ne, nch = int(layout[0]["n_elements"]), int(layout[0]["channels"])
    ps_sce = he and cc == 0 and asc is not None and (how != "asc_implicit" or nch == 1)
    slot_out = [2 if ps_sce and int(layout[0]["elem"][e]["type"]) == SCE else int(layout[0]["elem"][e]["channels"]) for e in range(ne)]
    first_out = [sum(slot_out[:e]) for e in range(ne)]
    nch = sum(slot_out)
    if asc is not None:
        # (implicit Parametric Stereo shows in the first access unit, not at open)
        assert ctx.channels == (1 if mode == "he_pce_ps_mono_implicit" else nch) and ctx.channel_layout == int(layout[0]["channel_layout"])
    st = np.zeros(pkg.MAX_ELEMENTS, pkg.AAC_STREAM_DT)
    length = 2048 if he and not down else 1024
    slot_ch = [int(layout[0]["elem"][e]["channels"]) for e in range(ne)]
    he_cfg = [pkg.CFG_HEV1 if c == 2 else (pkg.CFG_HEV2 if o == 2 else pkg.CFG_HEV1_MONO) for c, o in zip(slot_ch, slot_out)]
    state = [np.zeros((1, pkg.STATE_WORDS[h if he else (pkg.CFG_LC_STEREO if c == 2 else pkg.CFG_LC_MONO)]), np.float32)
             for c, h in zip(slot_ch, he_cfg)]
    pred = [np.tile(np.array([0, 0, 1, 1, 0, 0], np.float32), (1, c * pkg.MAX_PREDICTORS, 1)).reshape(1, -1) for c in slot_ch]
    ref_rng = np.full(1, 0x1f2e3d4c, np.int32)
    tab = pkg.SbrHeaderTable(64)
    sst = pkg.sbr_streams(ne)
    writers = {k: SW.SbrStreamWriter(pkg, 2 if t == CPE else 1, ps=ps_sce and t == SCE) for k, (t, _) in enumerate(elems)}
    ps_frames = 0
    # an LFE with SBR payloads of its own (mode he_5_1, frames 1 and 3): the reference's SBR reader takes their headers
    # and refuses the data (aacsbr.c:996-1000)
    lfe_payload = lambda t: mode == "he_5_1" and t in (1, 3)
    out = (C.c_int16 * (192000 // 2))()
    loud = lfe_headers = 0
    for t in range(5):
        payloads = None
        if he:
            payloads = []
            for k, (typ, _) in enumerate(elems):
                if typ == LFE and not lfe_payload(t):
                    payloads.append(None)
                    continue
                w = writers[k]
                while True:
                    keep = copy.deepcopy((w.ch, w.ps, w.header, w.hdr_rec, w.kx_m, w.coupling))
                    bits, _ = w.frame(rng, new_header=(t == 3 and k == 1) or typ == LFE, respec=(t == 3 and (k == 1 or typ == LFE)))
                    if (4 + len(bits) + 7) // 8 <= 269:
                        break
                    w.ch, w.ps, w.header, w.hdr_rec, w.kx_m, w.coupling = keep
                payloads.append(bits)
        lead = None
        if how == "adts" and cc == 0:
            # a data stream and a fill element may stand in front of the program config element (first frame)
            lead = lambda bw: (TL.write_dse(bw, rng) if t == 0 else None, TL.write_fill(bw, rng, 0x1, 3) if t == 0 else None,
                               bw.put(5, 3), bw.put(0, 4), TL.write_pce_body(bw, np.random.default_rng(1), *pce_elems))
        au, _ = TL.build(rng, si, aot, elems, extras=t & 1, payloads=payloads, lead=lead)
        pkt_bytes = _adts(au, aot, si, cc) if how == "adts" else au
        buf = C.create_string_buffer(pkt_bytes, len(pkt_bytes))
        pkt = HeaacPacket(C.cast(buf, C.c_void_p), len(pkt_bytes))
        size = C.c_int(192000)
        used = lib.heaac_codec_decode(C.byref(ctx), out, C.byref(size), C.byref(pkt))
        assert used == len(pkt_bytes), (t, used)
        assert (ctx.channels, ctx.frame_size, ctx.sample_rate) == (nch, length, 24000 if down else 48000), t
        assert ctx.channel_layout == int(layout[0]["channel_layout"]) and size.value == length * nch * 2
        got = np.frombuffer(out, np.int16, size.value // 2).reshape(length, nch).copy()
        # ---- the checker ----
        r, p = pkg.aac_parse_frame_layout(m4, layout, st, pkt_bytes)
        assert r == 0
        order = sorted(range(ne), key=lambda e: int(p["elem"][e]["seq"]))
        spec = [None] * ne
        for e in order:
            c = slot_ch[e]
            co = np.ascontiguousarray(p["coeffs"][e:e + 1, :c])
            if aot == 1:
                spec[e], ref_rng, pred[e] = oracle.spectral_tools_batch(c, co, p["tools"][e:e + 1], rng=ref_rng, pred=pred[e])
            else:
                spec[e], ref_rng = oracle.spectral_tools_batch(c, co, p["tools"][e:e + 1], rng=ref_rng)
        planes = [None] * nch
        for e in range(ne):
            c = slot_ch[e]
            ics = np.ascontiguousarray(p["ics"][e:e + 1, :c])
            if he:
                ei = p["elem"][e]
                with_ps = he_cfg[e] == pkg.CFG_HEV2
                ps_rec = None
                if int(ei["sbr_payload_bit"]) >= 0:
                    rr, sbr, ps_rec, _ = pkg.sbr_parse_payload(sst[e], tab, 24000, pkt_bytes, c, with_ps, bit=int(ei["sbr_payload_bit"]),
                                                          cnt=int(ei["sbr_payload_bytes"]), misplaced=bool(ei["sbr_misplaced"]))
                    assert rr == (-1 if ei["sbr_misplaced"] else 0)
                    assert bool(ei["sbr_misplaced"]) == (int(layout[0]["elem"][e]["type"]) == LFE)
                    if ei["sbr_misplaced"]:
                        assert int(sbr["start"][0]) == 0 and int(sbr["hdr"][0]) > 0
                        lfe_headers += 1
                else:
                    sbr = pkg.sbr_no_payload(sst[e], c)              # an LFE without a payload ("pure upsampling")
                    assert int(layout[0]["elem"][e]["type"]) == LFE
                ps_frames += int(with_ps and int(ps_rec["start"][0]) == 1)
                pcm, state[e] = oracle.he_decode_batch(he_cfg[e], spec[e], ics, sbr, tab.headers(), ps_rec if with_ps else None,
                                                       state[e], oracle.PCM_F32, downsampled=down)
            else:
                pcm, state[e] = oracle.lc_decode_batch(c, spec[e], ics, state[e], oracle.PCM_F32)
            for j in range(slot_out[e] if he else c):
                planes[first_out[e] + j] = pcm[0, j]
        want = oracle.float_to_int16_interleave(planes)
        assert np.array_equal(got, want), ("frame %d" % t, np.argwhere(got != want)[:4])
        loud = max(loud, int(np.abs(got.astype(int)).max()))
    assert loud > 50 and lfe_headers == (2 if mode == "he_5_1" else 0)
    assert (ps_frames > 0) == ps_sce
    # an access unit that leaves an element out is refused (the reference would transform stale buffers)
    au, _ = TL.build(rng, si, aot, elems[:-1], extras=False)
    pkt_bytes = _adts(au, aot, si, cc) if how == "adts" else au
    buf = C.create_string_buffer(pkt_bytes, len(pkt_bytes))
    pkt = HeaacPacket(C.cast(buf, C.c_void_p), len(pkt_bytes))
    size = C.c_int(192000)
    assert lib.heaac_codec_decode(C.byref(ctx), out, C.byref(size), C.byref(pkt)) < 0
    assert lib.heaac_codec_close(C.byref(ctx)) == 0
