"""What a refused access unit leaves behind (include/heaac_parse.h, HEAAC_REFUSED_*): the reference returns from the
middle of its element loop and undoes nothing (aacdec.c:2069-2070), so the window history, the noise generator and
the AAC-Main predictors of the stream have moved as far as its element decoders had got.  tests/refused_units.py
writes such units with a model of that state derived from what was written.  CPU: the parser's verdict, the history
it leaves in the stream and the records it hands the spectral tools.  GPU: streams through the pipeline and through
the codec surface carry on, after the refused unit, exactly like the call-by-call path driven by the model."""
import ctypes as C

import numpy as np
import pytest

import refused_units as R
import test_parse as TP

SI = 3


def _history(st, c):
    return int(st["window_sequence"][0, c]), int(st["use_kb_window"][0, c])


def _record_draws(t):
    """Numbers a tools channel record draws: the lines of its noise bands."""
    ics, total, idx = t["ics"], 0, 0
    for g in range(int(ics["num_window_groups"])):
        for i in range(int(ics["max_sfb"])):
            if int(t["band_type"][idx]) == 13:
                total += int(ics["group_len"][g]) * (int(ics["swb_offset"][i + 1]) - int(ics["swb_offset"][i]))
            idx += 1
    return total


def _units(kinds, writer, aot, seed, per_kind=6):
    rng = np.random.default_rng(seed)
    return [(k,) + writer(rng, SI, aot, k) for k in kinds for _ in range(per_kind)]


@pytest.mark.parametrize("aot", [2, 1])
@pytest.mark.parametrize("cpe", [False, True])
def test_the_parser_leaves_the_stream_where_the_reference_decoder_would_be(pkg, cpe, aot):
    cfg = TP._cfg(pkg, aot, SI, 2 if cpe else 1)
    seen = set()
    for kind, bad, good, model in _units(R.KINDS_CPE if cpe else R.KINDS_SCE, R.cpe_unit if cpe else R.sce_unit, aot, 77 + cpe + 2 * aot):
        st = np.zeros(1, pkg.AAC_STREAM_DT)
        st["window_sequence"][0] = (3, 1); st["use_kb_window"][0] = (1, 1)
        before = [_history(st, 0), _history(st, 1)]
        r, out = pkg.aac_parse_frame_ex(cfg, st[0:1], bad, with_cce=False)
        info = out["info"][0]
        assert r < 0 and info["channels"] == 0 and info["sbr_payload_bit"] == -1, kind
        if kind == "ours_only":
            assert info["refused"] == 0 and [_history(st, 0), _history(st, 1)] == before
            continue
        assert info["refused"] & pkg.REFUSED_AS_REFERENCE, kind
        for c in range(2):
            want = model["history"][c] if c < len(model["history"]) and model["history"][c] is not None else before[c]
            assert _history(st, c) == want, (kind, c)
        seen.add(kind)
        t = out["tools"][0]
        run = bool(info["refused"] & pkg.REFUSED_RUN_TOOLS)
        assert run == bool(model["draws"] or model["predicted"]), kind
        if not run:
            continue
        assert sum(_record_draws(t["ch"][c]) for c in range(2 if cpe else 1)) == model["draws"], kind
        for c in range(2 if cpe else 1):
            pr, ics = t["ch"][c]["pred"], t["ch"][c]["ics"]
            if aot == 1 and c not in model["predicted"]:
                # nothing the prediction stage could do to this channel: no line below the limit, no reset
                assert ics["num_windows"] == 1 and pr["pred_sfb_max"] == 0 and pr["predictor_reset_group"] == 0, kind
        if model["predicted"]:
            # the completed channels' records are the undamaged twin's, spectrum included
            st2 = np.zeros(1, pkg.AAC_STREAM_DT)
            r2, twin = pkg.aac_parse_frame_ex(cfg, st2[0:1], good, with_cce=False)
            assert r2 == 0
            for c in model["predicted"]:
                a, b = t["ch"][c], twin["tools"][0]["ch"][c]
                for f in ("ics", "band_type", "sf", "pred"):
                    assert a[f].tobytes() == b[f].tobytes(), (kind, c, f)
                assert np.array_equal(out["coeffs"][c].view(np.uint32), twin["coeffs"][c].view(np.uint32)), (kind, c)
    assert seen == set(R.KINDS_CPE if cpe else R.KINDS_SCE) - {"ours_only"}


def test_the_he_parser_passes_the_verdict_on(pkg):
    """heaac_heaac_parse_frame_ex: channels = 0 and the flags of the core parser."""
    import test_sbr_parse as TS
    rng = np.random.default_rng(5)
    bad, _, model = R.sce_unit(rng, SI, 2, "esc_overflow")
    m4 = TS._he_cfg(pkg, 1, True)
    m4.sampling_index, m4.sample_rate = SI, 48000
    st, sst, tab = np.zeros(1, pkg.AAC_STREAM_DT), pkg.sbr_streams(1), pkg.SbrHeaderTable(8)
    p = pkg.heaac_parse_batch(m4, st, sst, tab, [bad], with_ps=True)
    assert p["status"][0] < 0 and p["info"]["channels"][0] == 0
    assert p["info"]["refused"][0] == pkg.REFUSED_AS_REFERENCE | pkg.REFUSED_RUN_TOOLS
    assert _history(st, 0) == model["history"][0] and _record_draws(p["tools"][0]["ch"][0]) == model["draws"] > 0


# ------------------------------------------------------------------------------------------------------------------
# GPU: streams carry on behind a refused unit as the model says
# ------------------------------------------------------------------------------------------------------------------
def _expected_run(pkg, oracle, cfg, ch, aot, ticks_good, bad_tick, models):
    """The oracle over the good units, the refused tick replaced per stream by its model: no samples and the
    decoder's overlap state kept; the window history set; the generator stepped `draws` times; the predictors of the
    completed channels stepped by the oracle's tools on the undamaged twin's records."""
    n = len(ticks_good[0])
    st = np.zeros(n, pkg.AAC_STREAM_DT)
    state = np.zeros((n, 512 * ch), np.float32)
    rs = np.full(n, 0x1f2e3d4c, np.int32)
    pred = np.tile(np.array([0, 0, 1, 1, 0, 0], np.float32), (n, ch, pkg.MAX_PREDICTORS, 1)) if aot == 1 else None
    out = []
    for t, aus in enumerate(ticks_good):
        before = st.copy()
        q = pkg.aac_parse_batch(cfg, st, aus, threads=1)
        assert q["failed"] == 0
        coeffs = np.ascontiguousarray(q["coeffs"][:, :ch])
        tools = q["tools"].copy()
        rs_before = rs.copy()
        if t == bad_tick:
            for i, m in models.items():
                st[i] = before[i]
                for c, h in enumerate(m["history"]):
                    if h is not None:
                        st["window_sequence"][i, c], st["use_kb_window"][i, c] = h
                # the tools see the twin's records for the channels that were predicted and nothing for the others
                for c in range(2):
                    if c not in m["predicted"]:
                        tools["ch"][i, c] = np.zeros((), tools.dtype["ch"].base)
                tools["common_window"][i] = tools["ms_present"][i] = 0
                tools["ch"]["tns"]["present"][i] = 0
        if aot == 1:
            ref_c, rs, pred = oracle.spectral_tools_batch(ch, coeffs, tools, rng=rs, pred=pred)
        else:
            ref_c, rs = oracle.spectral_tools_batch(ch, coeffs, tools, rng=rs)
        if t == bad_tick:
            for i, m in models.items():
                stepped = R.lcg(rs_before[i], m["draws"])
                # (a predicted channel's numbers were drawn by its own records: the same count)
                assert not m["predicted"] or rs[i] == stepped, (i, m)
                rs[i] = stepped
        ics = np.ascontiguousarray(q["ics"][:, :ch])
        pcm, new_state = oracle.lc_decode_batch(ch, ref_c, ics, state, oracle.PCM_S16)
        pcm = pcm.copy()
        if t == bad_tick:
            for i in models:
                new_state[i] = state[i]
                pcm[i] = 0
        state = new_state
        out.append(pcm)
    return out


def _streams(pkg, cpe, aot, seed, ticks=5, bad_tick=2, rounds=2):
    rng = np.random.default_rng(seed)
    kinds = R.KINDS_CPE if cpe else R.KINDS_SCE
    writer = R.cpe_unit if cpe else R.sce_unit
    n = rounds * len(kinds) + 3
    good = [[TP._write_au(rng, SI, aot, cpe, extras=False, quiet=True)[0] for _ in range(n)] for _ in range(ticks)]
    fed = [list(a) for a in good]
    models = {}
    for j, kind in enumerate(kinds * rounds):
        i = j + 1
        bad, twin, model = writer(rng, SI, aot, kind)
        fed[bad_tick][i] = bad
        good[bad_tick][i] = twin
        models[i] = model
    return good, fed, models, n


@pytest.mark.gpu
@pytest.mark.parametrize("aot", [2, 1])
@pytest.mark.parametrize("cpe", [False, True])
def test_streams_carry_on_behind_a_refused_unit_as_the_reference_decoder_would(pkg, oracle, dev, cpe, aot):
    """Through the pipeline: a batch in which every kind of refused unit hits some stream at one tick."""
    ch, bad_tick = 2 if cpe else 1, 2
    cfg = TP._cfg(pkg, aot, SI, ch)
    good, fed, models, n = _streams(pkg, cpe, aot, 31 + cpe + 2 * aot, bad_tick=bad_tick)
    want = _expected_run(pkg, oracle, cfg, ch, aot, good, bad_tick, models)
    pl = pkg.Pipeline(cfg, pkg.CFG_LC_STEREO if cpe else pkg.CFG_LC_MONO, n, threads=2)
    for t in range(len(fed)):
        status = pl.submit(fed[t])
        got = pl.collect().copy()
        assert [i for i in range(n) if status[i] < 0] == (sorted(models) if t == bad_tick else []), t
        assert np.array_equal(got, want[t]), (t, [i for i in range(n) if not np.array_equal(got[i], want[t][i])])
    pl.close()
    assert all(any(want[t][i].any() for t in range(bad_tick + 1, len(fed))) for i in models)
    # the model matters: a stream that simply skipped the unit would sound different afterwards
    skipped = _expected_run(pkg, oracle, cfg, ch, aot, good, bad_tick,
                            {i: dict(history=[None, None], draws=0, predicted=[]) for i in models})
    differ = [i for i in models if any(not np.array_equal(skipped[t][i], want[t][i]) for t in range(bad_tick + 1, len(fed)))]
    assert len(differ) >= len(models) // 2, differ


@pytest.mark.gpu
@pytest.mark.parametrize("aot", [2, 1])
@pytest.mark.parametrize("cpe", [False, True])
def test_the_codec_surface_carries_on_behind_a_refused_unit(pkg, oracle, dev, cpe, aot):
    """heaac_codec_decode, one stream per kind: -1 and no samples for the refused unit, then on as the model says."""
    from test_shim_gpu import HeaacCodecContext, HeaacPacket
    lib = pkg.lib()
    ch, bad_tick = 2 if cpe else 1, 2
    cfg = TP._cfg(pkg, aot, SI, ch)
    good, fed, models, n = _streams(pkg, cpe, aot, 57 + cpe + 2 * aot, bad_tick=bad_tick, rounds=1)
    want = _expected_run(pkg, oracle, cfg, ch, aot, good, bad_tick, models)
    asc = bytes([(aot << 3) | (SI >> 1), ((SI & 1) << 7) | (ch << 3)])
    codec = C.c_void_p.in_dll(lib, "heaac_aac_decoder")
    out = (C.c_int16 * (192000 // 2))()
    for i in range(n):
        ctx = HeaacCodecContext(cfg=-1, extradata=asc, extradata_size=len(asc))
        assert lib.heaac_codec_open(C.byref(ctx), C.c_void_p(C.addressof(codec))) == 0
        for t in range(len(fed)):
            buf = C.create_string_buffer(fed[t][i], len(fed[t][i]))
            pkt = HeaacPacket(C.cast(buf, C.c_void_p), len(fed[t][i]))
            size = C.c_int(192000)
            used = lib.heaac_codec_decode(C.byref(ctx), out, C.byref(size), C.byref(pkt))
            if t == bad_tick and i in models:
                assert used < 0, (i, t)
                continue
            assert used == len(fed[t][i]) and size.value == 1024 * ch * 2, (i, t, used)
            got = np.frombuffer(out, np.int16, 1024 * ch).reshape(1024, ch)
            assert np.array_equal(got, want[t][i]), (i, t)
        assert lib.heaac_codec_close(C.byref(ctx)) == 0


# ------------------------------------------------------------------------------------------------------------------
# several output elements per unit (a 3.0 layout: centre SCE, then the front pair)
# ------------------------------------------------------------------------------------------------------------------
def _layout_3_0(pkg, aot):
    cfg = TP._cfg(pkg, aot, SI, 3)
    r, layout = pkg.aac_layout_default(3)
    assert r == 0 and int(layout[0]["n_elements"]) == 2
    # slot of the centre element and of the pair in the layout's (output) order
    slot = {int(layout[0]["elem"][e]["channels"]): e for e in range(2)}
    return cfg, layout, slot[1], slot[2]


@pytest.mark.parametrize("aot", [2, 1])
def test_the_layout_parser_leaves_every_element_where_the_reference_decoder_would_be(pkg, aot):
    cfg, layout0, s_sce, s_cpe = _layout_3_0(pkg, aot)
    rng = np.random.default_rng(311 + aot)
    for kind in R.KINDS_3_0 * 4:
        bad, twin, model = R.unit_3_0(rng, SI, aot, kind)
        layout = layout0.copy()
        st = np.zeros(pkg.MAX_ELEMENTS, pkg.AAC_STREAM_DT)
        st["window_sequence"][:2] = (3, 1); st["use_kb_window"][:2] = (1, 1)
        before = st.copy()
        r, out = pkg.aac_parse_frame_layout(cfg, layout, st, bad)
        info = out["info"][0]
        assert r < 0 and info["channels"] == 0, kind
        if kind == "ours_only":
            assert info["refused"] == 0 and st.tobytes() == before.tobytes()
            continue
        assert info["refused"] & pkg.REFUSED_AS_REFERENCE, kind
        draws = 0
        for e, m, nch in ((s_sce, model["elements"][0], 1), (s_cpe, model["elements"][1], 2)):
            for c in range(nch):
                want = m["history"][c] if m is not None and m["history"][c] is not None else (int(before["window_sequence"][e, c]), int(before["use_kb_window"][e, c]))
                assert (int(st["window_sequence"][e, c]), int(st["use_kb_window"][e, c])) == want, (kind, e, c)
            if out["elem"][e]["present"]:
                draws += sum(_record_draws(out["tools"][e]["ch"][c]) for c in range(nch))
            else:
                assert m is None or not (m["draws"] or m["predicted"]), kind
        run = bool(info["refused"] & pkg.REFUSED_RUN_TOOLS)
        assert run == bool(model["draws"] or any(m and m["predicted"] for m in model["elements"])), kind
        if run:
            assert draws == model["draws"], kind
            # bitstream order: the centre element first
            assert int(out["elem"][s_sce]["seq"]) == 0 and (not out["elem"][s_cpe]["present"] or int(out["elem"][s_cpe]["seq"]) == 1)


def _expected_layout_run(pkg, oracle, cfg, layout0, slots, aot, ticks_good, bad_tick, models):
    """The oracle over the good units of n 3.0 streams (tools through the elements in bitstream order on one
    generator per stream, decode per element on its own state, float_to_int16_interleave over the planes in layout
    order), the refused tick replaced per stream by its model as in _expected_run."""
    n = len(ticks_good[0])
    s_sce, s_cpe = slots
    nch = {s_sce: 1, s_cpe: 2}
    first = {e: int(layout0[0]["elem"][e]["first_channel"]) for e in (s_sce, s_cpe)}
    layouts = [layout0.copy() for _ in range(n)]
    st = [np.zeros(pkg.MAX_ELEMENTS, pkg.AAC_STREAM_DT) for _ in range(n)]
    state = {e: np.zeros((n, 512 * nch[e]), np.float32) for e in nch}
    pred = {e: np.tile(np.array([0, 0, 1, 1, 0, 0], np.float32), (n, nch[e], pkg.MAX_PREDICTORS, 1)) for e in nch}
    rs = np.full(n, 0x1f2e3d4c, np.int32)
    out = []
    for t, aus in enumerate(ticks_good):
        pcm_t = np.zeros((n, 1024, 3), np.int16)
        for i in range(n):
            before = st[i].copy()
            r, p = pkg.aac_parse_frame_layout(cfg, layouts[i], st[i], aus[i])
            assert r == 0
            m = models.get(i) if t == bad_tick else None
            rs_before = int(rs[i])
            planes = [None] * 3
            for e in (s_sce, s_cpe):                       # bitstream order of a 3.0 unit
                c = nch[e]
                tools = p["tools"][e:e + 1].copy()
                if m is not None:
                    em = m["elements"][0 if e == s_sce else 1]
                    for k in range(2):
                        if em is None or k not in em["predicted"]:
                            tools["ch"][0, k] = np.zeros((), tools.dtype["ch"].base)
                    tools["common_window"][0] = tools["ms_present"][0] = 0
                    tools["ch"]["tns"]["present"][0] = 0
                co = np.ascontiguousarray(p["coeffs"][e:e + 1, :c])
                if aot == 1:
                    spec, r1, pr = oracle.spectral_tools_batch(c, co, tools, rng=rs[i:i + 1], pred=pred[e][i:i + 1])
                    pred[e][i:i + 1] = pr
                else:
                    spec, r1 = oracle.spectral_tools_batch(c, co, tools, rng=rs[i:i + 1])
                rs[i] = r1[0]
                if m is not None:
                    continue
                ics = np.ascontiguousarray(p["ics"][e:e + 1, :c])
                pcm, state[e][i:i + 1] = oracle.lc_decode_batch(c, spec, ics, state[e][i:i + 1], oracle.PCM_F32)
                for j in range(c):
                    planes[first[e] + j] = pcm[0, j]
            if m is not None:
                st[i][:] = before
                for e, em in ((s_sce, m["elements"][0]), (s_cpe, m["elements"][1])):
                    for c, h in enumerate(em["history"] if em is not None else []):
                        if h is not None:
                            st[i]["window_sequence"][e, c], st[i]["use_kb_window"][e, c] = h
                rs[i] = R.lcg(rs_before, m["draws"])
            else:
                pcm_t[i] = oracle.float_to_int16_interleave(planes)
        out.append(pcm_t)
    return out


def _streams_3_0(pkg, aot, seed, ticks=5, bad_tick=2, rounds=2):
    import test_parse_layout as TL
    rng = np.random.default_rng(seed)
    n = rounds * len(R.KINDS_3_0) + 2
    good = [[TL.build(rng, SI, aot, [(0, 0), (1, 0)], extras=False)[0] for _ in range(n)] for _ in range(ticks)]
    fed = [list(a) for a in good]
    models = {}
    for j, kind in enumerate(R.KINDS_3_0 * rounds):
        bad, twin, model = R.unit_3_0(rng, SI, aot, kind)
        fed[bad_tick][j + 1] = bad
        good[bad_tick][j + 1] = twin
        models[j + 1] = model
    return good, fed, models, n


@pytest.mark.gpu
@pytest.mark.parametrize("aot", [2, 1])
def test_layout_streams_carry_on_behind_a_refused_unit(pkg, oracle, dev, aot):
    """3.0 streams through the layout pipeline and through one codec context each."""
    from test_shim_gpu import HeaacCodecContext, HeaacPacket
    lib = pkg.lib()
    bad_tick = 2
    cfg, layout, s_sce, s_cpe = _layout_3_0(pkg, aot)
    cfg.sbr = 0
    good, fed, models, n = _streams_3_0(pkg, aot, 91 + aot, bad_tick=bad_tick)
    want = _expected_layout_run(pkg, oracle, cfg, layout, (s_sce, s_cpe), aot, good, bad_tick, models)
    pl = pkg.LayoutPipeline(cfg, layout, n, threads=2)
    for t in range(len(fed)):
        status = pl.submit(fed[t])
        got = pl.collect().copy()
        assert [i for i in range(n) if status[i] < 0] == (sorted(models) if t == bad_tick else []), t
        assert np.array_equal(got, want[t]), (t, [i for i in range(n) if not np.array_equal(got[i], want[t][i])])
    pl.close()
    asc = bytes([(aot << 3) | (SI >> 1), ((SI & 1) << 7) | (3 << 3)])
    codec = C.c_void_p.in_dll(lib, "heaac_aac_decoder")
    out = (C.c_int16 * (192000 // 2))()
    for i in sorted(models):
        ctx = HeaacCodecContext(cfg=-1, extradata=asc, extradata_size=len(asc))
        assert lib.heaac_codec_open(C.byref(ctx), C.c_void_p(C.addressof(codec))) == 0
        for t in range(len(fed)):
            buf = C.create_string_buffer(fed[t][i], len(fed[t][i]))
            pkt = HeaacPacket(C.cast(buf, C.c_void_p), len(fed[t][i]))
            size = C.c_int(192000)
            used = lib.heaac_codec_decode(C.byref(ctx), out, C.byref(size), C.byref(pkt))
            if t == bad_tick:
                assert used < 0, (i, t)
                continue
            assert used == len(fed[t][i]) and size.value == 1024 * 3 * 2, (i, t, used)
            assert np.array_equal(np.frombuffer(out, np.int16, 1024 * 3).reshape(1024, 3), want[t][i]), (i, t)
        assert lib.heaac_codec_close(C.byref(ctx)) == 0
    # the model matters
    skipped = _expected_layout_run(pkg, oracle, cfg, layout, (s_sce, s_cpe), aot, good, bad_tick,
                                   {i: dict(elements=[None, None], draws=0) for i in models})
    differ = [i for i in models if any(not np.array_equal(skipped[t][i], want[t][i]) for t in range(bad_tick + 1, len(fed)))]
    assert len(differ) >= len(models) // 2, differ


# ------------------------------------------------------------------------------------------------------------------
# HE-AAC: the core element's refusal moves the core's state; the SBR / PS side stays where it was
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("cfgname", ["CFG_HEV2", "CFG_HEV1"])
def test_he_streams_carry_on_behind_a_refused_unit(pkg, oracle, dev, cfgname):
    """HE-AACv2 (mono + SBR + PS) and HE-AACv1 stereo streams through the pipeline: a unit refused inside its core
    element leaves the window history and the noise generator as the model says, the SBR / PS reader and the whole
    DSP state record as they were (ff_sbr_apply never ran: spectral_to_sample is not reached, aacdec.c:2069-2078)."""
    import test_pipeline as TPL
    import test_sbr_parse as TS
    hcfg = getattr(pkg, cfgname)
    cpe = hcfg == pkg.CFG_HEV1
    ch, ps, si, bad_tick, ticks = (2 if cpe else 1), not cpe, 6, 2, 5
    rng = np.random.default_rng(77 + cpe)
    kinds = [k for k in (R.KINDS_CPE if cpe else R.KINDS_SCE)]
    n = len(kinds) + 2
    good = TPL._ticks(pkg, rng, ch, ps, n, ticks)
    fed = [list(a) for a in good]
    models = {}
    for j, kind in enumerate(kinds):
        bad, _, model = (R.cpe_unit if cpe else R.sce_unit)(rng, si, 2, kind)
        fed[bad_tick][j + 1] = bad
        models[j + 1] = model
    m4 = TS._he_cfg(pkg, ch, ps)
    # the oracle over the units each stream decodes (a refused unit is absent from its stream's SBR / DSP history)
    tab = pkg.SbrHeaderTable(256)
    st, sst = np.zeros(n, pkg.AAC_STREAM_DT), pkg.sbr_streams(n)
    state = np.zeros((n, pkg.STATE_WORDS[hcfg]), np.float32)
    rs = np.full(n, 0x1f2e3d4c, np.int32)
    want = []
    for t in range(ticks):
        if t == bad_tick:
            # parse the good units of the other streams; the refused streams keep parser and decoder state
            idx = [i for i in range(n) if i not in models]
        else:
            idx = list(range(n))
        sub_st, sub_sst = st[idx].copy(), np.ascontiguousarray(sst[idx])
        p = pkg.heaac_parse_batch(m4, sub_st, sub_sst, tab, [good[t][i] for i in idx], with_ps=ps)
        # (behind the refused unit a stream's SBR payload may fail -- its time-differential data build on a frame the
        # reader never saw -- which turns SBR off for that unit on both sides; the core element always parses)
        assert (p["info"]["channels"] == ch).all()
        st[idx], sst[idx] = sub_st, sub_sst
        coeffs = np.ascontiguousarray(p["coeffs"][:, :ch])
        ref_c, r1 = oracle.spectral_tools_batch(ch, coeffs, p["tools"], rng=rs[idx])
        rs[idx] = r1
        pcm, s1 = oracle.he_decode_batch(hcfg, ref_c, np.ascontiguousarray(p["ics"][:, :ch]), p["sbr"], tab.headers(),
                                         p["ps"] if ps else None, state[idx], oracle.PCM_S16)
        state[idx] = s1
        full = np.zeros((n,) + pcm.shape[1:], np.int16)
        full[idx] = pcm
        want.append(full)
        if t == bad_tick:
            for i, m in models.items():
                for c, h in enumerate(m["history"]):
                    if h is not None:
                        st["window_sequence"][i, c], st["use_kb_window"][i, c] = h
                rs[i] = R.lcg(rs[i], m["draws"])
    pl = pkg.Pipeline(m4, hcfg, n, threads=2)
    for t in range(ticks):
        status = pl.submit(fed[t])
        got = pl.collect().copy()
        if t == bad_tick:
            assert all(status[i] < 0 for i in models) and all(status[i] == 0 for i in range(n) if i not in models)
        assert np.array_equal(got, want[t]), (t, [i for i in range(n) if not np.array_equal(got[i], want[t][i])])
    pl.close()
    assert all(any(want[t][i].any() for t in range(bad_tick + 1, ticks)) for i in models)


# ------------------------------------------------------------------------------------------------------------------
# get_che: which element a channel configuration 1 / 2 stream has, and under which tag
# ------------------------------------------------------------------------------------------------------------------
def test_elements_the_configuration_has_no_place_for_fail_the_unit_where_get_che_does(pkg):
    """aacdec.c:113-177, :2011-2015.  The one element of the configuration is mapped to the tag it is first met with:
    the other element type, the element under another tag later on, and a second element in one unit all find nothing
    allocated -- the unit fails there, with whatever stood in front of the refusal decoded."""
    rng = np.random.default_rng(12)
    # the wrong type: nothing is read, nothing moves, no tag is mapped
    for ch, writer in ((2, R.sce_element), (1, R.cpe_element)):
        cfg = TP._cfg(pkg, 2, SI, ch)
        st = np.zeros(1, pkg.AAC_STREAM_DT)
        st["window_sequence"][0] = (3, 1)
        before = st.copy()
        bits, _, _ = writer(rng, SI, 2, "good")
        r, out = pkg.aac_parse_frame_ex(cfg, st[0:1], R._bytes(bits + R.END), with_cce=False)
        assert r < 0 and out["info"][0]["refused"] == pkg.REFUSED_AS_REFERENCE and st.tobytes() == before.tobytes()
    # the tag: the first one met stays
    cfg = TP._cfg(pkg, 2, SI, 1)
    st = np.zeros(1, pkg.AAC_STREAM_DT)
    for tag, ok in ((5, True), (5, True), (0, False), (6, False), (5, True)):
        bits, _, model = R.sce_element(rng, SI, 2, "good", tag=tag)
        before = st.copy()
        r, out = pkg.aac_parse_frame_ex(cfg, st[0:1], R._bytes(bits + R.END), with_cce=False)
        assert (r == 0) == ok, (tag, r)
        assert int(st["mapped_tag"][0]) == 6
        if ok:
            assert int(out["info"][0]["elem_id"]) == 5 and _history(st, 0) == model["history"][0]
        else:
            assert out["info"][0]["refused"] == pkg.REFUSED_AS_REFERENCE and st.tobytes() == before.tobytes()
    # a second element in the unit: refused behind the first, whose decoder has run
    for ch, writer in ((1, R.sce_element), (2, R.cpe_element)):
        cfg = TP._cfg(pkg, 2, SI, ch)
        for second in (R.sce_element, R.cpe_element):
            st = np.zeros(1, pkg.AAC_STREAM_DT)
            a, _, model = writer(rng, SI, 2, "good")
            b, _, _ = second(rng, SI, 2, "good", tag=1)
            r, out = pkg.aac_parse_frame_ex(cfg, st[0:1], R._bytes(a + b + R.END), with_cce=False)
            info = out["info"][0]
            assert r < 0 and info["refused"] & pkg.REFUSED_AS_REFERENCE
            assert [_history(st, c) for c in range(ch)] == model["history"]
            assert bool(info["refused"] & pkg.REFUSED_RUN_TOOLS) == bool(model["draws"])
            if model["draws"]:
                assert sum(_record_draws(out["tools"][0]["ch"][c]) for c in range(ch)) == model["draws"]
