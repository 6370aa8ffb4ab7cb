"""include/heaac_pipeline.h: the overlapped host-buffer pipeline.  CPU: argument checks and the loud failure
without a device.  GPU: several ticks of HE-AACv2 / HE-AACv1 streams through the pipeline give, tick for tick and
bit for bit, the PCM of the same access units through the batch parser + spectral tools + decode made one call
after the other (the path the other tests pin to the oracle)."""
import ctypes as C
import copy

import numpy as np
import pytest

import sbr_bitwriter as SW
import test_parse as TP
import test_sbr_parse as TS


def test_create_checks_arguments_and_fails_loudly_without_a_device(pkg):
    import torch
    lib = pkg.lib()
    h = C.c_void_p(1)
    cfg = TS._he_cfg(pkg, 1, True)
    assert lib.heaac_pipeline_create(C.byref(h), C.byref(cfg), 99, C.c_size_t(4), 1) == -1 and not h.value
    assert lib.heaac_pipeline_create(C.byref(h), None, pkg.CFG_HEV2, C.c_size_t(4), 1) == -1
    assert lib.heaac_pipeline_create(C.byref(h), C.byref(cfg), pkg.CFG_HEV2, C.c_size_t(0), 1) == -1
    if not torch.cuda.is_available():
        assert lib.heaac_pipeline_create(C.byref(h), C.byref(cfg), pkg.CFG_HEV2, C.c_size_t(4), 1) == -4 and not h.value
    # the layout form: argument checks first, then the same loud failure
    r, layout = pkg.aac_layout_default(6)
    m4 = TP._cfg(pkg, 2, 3, 6)
    lp = lambda c, l, n: lib.heaac_layout_pipeline_create(C.byref(h), C.byref(c) if c is not None else None,
                                                          l.ctypes.data_as(C.c_void_p) if l is not None else None, C.c_size_t(n), 1)
    assert r == 0 and lp(None, layout, 4) == -1 and lp(m4, None, 4) == -1 and lp(m4, layout, 0) == -1
    m4.sbr = -1
    assert lp(m4, layout, 4) == -1                          # implicit SBR is settled per stream
    m4.sbr = 0
    if not torch.cuda.is_available():
        assert lp(m4, layout, 4) == -4 and not h.value


def _ticks(pkg, rng, channels, ps, n, ticks):
    writers = [SW.SbrStreamWriter(pkg, channels, ps=ps, ps_modes="20" if ps else "any") for _ in range(n)]
    out = []
    for t in range(ticks):
        aus = []
        for w in writers:
            while True:
                keep = copy.deepcopy((w.ch, w.ps, w.header, w.hdr_rec, w.kx_m, w.coupling))
                bits, _ = w.frame(rng, new_header=(t == 3))
                if (4 + len(bits) + 7) // 8 <= 269:
                    break
                w.ch, w.ps, w.header, w.hdr_rec, w.kx_m, w.coupling = keep
            aus.append(TP._write_au(rng, 6, 2, channels == 2, extras=False, sbr=(bits, False), quiet=True)[0])
        out.append(aus)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("aot", [2, 1])
def test_pipeline_on_plain_aac_lc_streams(pkg, dev, aot):
    """AAC-LC / AAC-Main stereo streams (no SBR): parse + tools + heaac_lc_decode_batch per tick; the Main streams'
    predictor state lives in the pipeline."""
    import torch
    rng = np.random.default_rng(8 + aot)
    n, ticks, si = 29, 6, 3
    aus = [[TP._write_au(rng, si, aot, True, extras=True, quiet=True)[0] for _ in range(n)] for _ in range(ticks)]
    cfg = TP._cfg(pkg, aot, si, 2)
    pl = pkg.Pipeline(cfg, pkg.CFG_LC_STEREO, n, threads=2)
    st = np.zeros(n, pkg.AAC_STREAM_DT)
    d_state = torch.zeros((n, 1024), device="cuda")
    d_rng = torch.full((n,), 0x1f2e3d4c, dtype=torch.int32, device="cuda")
    d_pred = torch.tensor([0, 0, 1, 1, 0, 0], dtype=torch.float32, device="cuda").repeat(n, 2 * pkg.MAX_PREDICTORS, 1) if aot == 1 else None
    predicted = 0
    got = []
    for t in range(ticks):
        pl.submit(aus[t])
        if t >= 2:
            got.append(pl.collect().copy())
    while len(got) < ticks:
        got.append(pl.collect().copy())
    for t in range(ticks):
        q = pkg.aac_parse_batch(cfg, st, aus[t], threads=1)
        assert q["failed"] == 0
        coeffs = torch.from_numpy(q["coeffs"]).cuda()
        dev.spectral_tools(2, coeffs, pkg.to_device(q["tools"]), rng=d_rng, pred=d_pred)
        predicted += int(q["tools"]["ch"]["pred"]["predictor_present"].sum())
        pcm, d_state = dev.lc_decode(2, coeffs, pkg.to_device(q["ics"]), d_state, pcm_format=pkg.PCM_S16)
        torch.cuda.synchronize()
        assert got[t].shape == (n, 1024, 2) and np.array_equal(got[t], pcm.cpu().numpy()), t
    assert (predicted > 20) == (aot == 1)
    pl.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfgname", ["CFG_HEV2", "CFG_HEV1", "CFG_HEV1:downsampled"])
def test_pipeline_equals_the_calls_made_one_after_the_other(pkg, dev, cfgname):
    import torch
    cfgname, _, down = cfgname.partition(":")
    cfg = getattr(pkg, cfgname)
    ps = cfg == pkg.CFG_HEV2
    channels = 1 if ps else 2
    rng = np.random.default_rng(61 + channels)
    n, ticks = 37, 7
    aus = _ticks(pkg, rng, channels, ps, n, ticks)
    m4 = TS._he_cfg(pkg, channels, ps)
    if down:                                               # extension rate = core rate: 1024 samples per tick (aacsbr.c:1719)
        m4.ext_sampling_index, m4.ext_sample_rate = m4.sampling_index, m4.sample_rate
    pl = pkg.Pipeline(m4, cfg, n, threads=3)
    # the direct path
    tab = pkg.SbrHeaderTable(4096)
    st, sst = np.zeros(n, pkg.AAC_STREAM_DT), pkg.sbr_streams(n)
    d_state = torch.zeros((n, pkg.STATE_WORDS[cfg]), device="cuda")
    d_rng = torch.full((n,), 0x1f2e3d4c, dtype=torch.int32, device="cuda")
    want = []
    for t in range(ticks):
        q = pkg.heaac_parse_batch(m4, st, sst, tab, aus[t], threads=2, with_ps=ps)
        coeffs = torch.from_numpy(np.ascontiguousarray(q["coeffs"][:, :channels])).cuda()
        dev.spectral_tools(channels, coeffs, pkg.to_device(q["tools"]), rng=d_rng)
        pcm, d_state = dev.he_decode(cfg, coeffs, pkg.to_device(np.ascontiguousarray(q["ics"][:, :channels])),
                                     pkg.to_device(q["sbr"]), pkg.to_device(tab.headers()),
                                     pkg.to_device(q["ps"]) if ps else None, d_state, state_out=d_state, pcm_format=pkg.PCM_S16,
                                     **({"downsampled": True} if down else {}))
        torch.cuda.synchronize()
        want.append((pcm.cpu().numpy(), q["status"].copy()))
    # the pipeline, as many ticks in flight as it takes
    got, status = [], []
    depth = 4
    for t in range(ticks):
        if t >= depth:
            got.append(pl.collect().copy())
        status.append(pl.submit(aus[t]))
    while len(got) < ticks:
        got.append(pl.collect().copy())
    with pytest.raises(pkg.HeaacError):
        pl.collect()                                                   # nothing in flight
    loud = 0
    for t in range(ticks):
        assert np.array_equal(status[t], want[t][1]), t
        assert got[t].shape == (n, 1024 if down else 2048, 2) and np.array_equal(got[t], want[t][0]), "tick %d" % t
        loud = max(loud, int(np.abs(got[t].astype(int)).max()))
    assert loud > 50
    # one submit more than the depth without a collect is refused
    for t in range(depth):
        pl.submit(aus[t])
    with pytest.raises(pkg.HeaacError):
        pl.submit(aus[depth])
    for t in range(depth):
        pl.collect()
    tm = pl.timing()
    assert tm["parse"] > 0 and tm["gpu"] > 0
    pl.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfgname", ["CFG_HEV2", "CFG_LC_STEREO"])
def test_a_damaged_access_unit_gives_silence_and_leaves_its_stream_as_it_was(pkg, dev, cfgname):
    """one unit of one stream does not parse.  That
    stream's PCM of the tick is zero, and from the next tick on it decodes exactly as a stream that never saw the
    damaged unit; every other stream is untouched.  With and without a status array."""
    import torch
    cfg = getattr(pkg, cfgname)
    he = cfg == pkg.CFG_HEV2
    rng = np.random.default_rng(404 + he)
    n, ticks, bad_tick, bad = 9, 6, 2, [3, 7]
    if he:
        aus = _ticks(pkg, rng, 1, True, n, ticks)
        m4 = TS._he_cfg(pkg, 1, True)
    else:
        aus = [[TP._write_au(rng, 3, 1, True, extras=True, quiet=True)[0] for _ in range(n)] for _ in range(ticks)]
        m4 = TP._cfg(pkg, 1, 3, 2)                                   # AAC-Main: the predictors are state too
    good = [list(a) for a in aus]
    for i in bad:
        # a unit this parser alone refuses (a coupling element in a one-element stream): nothing of the stream moves.
        # (Where the refusal is the reference's own, the stream is left where ITS decoder would be:
        # tests/test_refused_units.py.)
        aus[bad_tick][i] = bytes([0x40, 0]) + bytes(8)
    results = []
    for with_status in (True, False):
        pl = pkg.Pipeline(m4, cfg, n, threads=2)
        got = []
        for t in range(ticks):
            st = pl.submit(aus[t], with_status=with_status)
            if t == bad_tick and with_status:
                assert all(st[i] < 0 for i in bad) and all(st[i] >= 0 for i in range(n) if i not in bad)
            got.append(pl.collect().copy())
        pl.close()
        results.append(got)
    assert all(np.array_equal(a, b) for a, b in zip(*results))
    got = results[0]
    # the reference run: the damaged tick is simply absent for the damaged streams
    pl = pkg.Pipeline(m4, cfg, n, threads=1)
    ref = []
    for t in range(ticks):
        pl.submit(good[t])
        ref.append(pl.collect().copy())
    pl.close()
    for i in range(n):
        if i not in bad:
            for t in range(ticks):
                assert np.array_equal(got[t][i], ref[t][i]), (t, i)
    # a damaged stream: silence at the tick, and afterwards what a pipeline gives that was fed the same units with
    # the damaged one left out
    for i in bad:
        assert not got[bad_tick][i].any()
        assert np.array_equal(got[bad_tick - 1][i], ref[bad_tick - 1][i])
    pl = pkg.Pipeline(m4, cfg, n, threads=1)
    skip = []
    for t in range(ticks):
        if t == bad_tick:
            continue
        pl.submit(good[t])
        skip.append(pl.collect().copy())
    pl.close()
    for i in bad:
        for t in range(bad_tick + 1, ticks):
            assert np.array_equal(got[t][i], skip[t - 1][i]), (t, i)
        assert any(got[t][i].any() for t in range(bad_tick + 1, ticks))
