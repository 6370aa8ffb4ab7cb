"""GPU parity: SBR filterbanks and the whole HE-AAC channel-element path (through the
C ABI) against the oracle.  Bar: bit-exact float PCM and state, identical int16."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _synth():
    return importlib.import_module("ffmpeg_heaac_amd.synth")


def _mismatch(a, b):
    d = _bits(a) != _bits(b)
    return int(d.sum()), (np.argwhere(d)[:5].tolist() if d.any() else [])


@pytest.mark.parametrize("n", [1, 9, 130])
def test_qmf_analysis_batch(pkg, oracle, dev, n):
    import torch
    rng = np.random.default_rng(n)
    x = (rng.standard_normal((n, 1024)) * 1e-3).astype(np.float32)
    xh = (rng.standard_normal((n, 288)) * 30).astype(np.float32)
    W, xo = dev.qmf_analysis(torch.from_numpy(x).cuda(), torch.from_numpy(xh).cuda())
    W, xo = W.cpu().numpy(), xo.cpu().numpy()
    for i in range(n):
        rW, rx = oracle.qmf_analysis(x[i], xh[i])
        assert np.array_equal(_bits(W[i]), _bits(rW)), i
        assert np.array_equal(_bits(xo[i]), _bits(rx)), i


@pytest.mark.parametrize("n", [1, 7, 100])
def test_qmf_synthesis_batch(pkg, oracle, dev, n):
    import torch
    rng = np.random.default_rng(50 + n)
    X = (rng.standard_normal((n, 2, 32, 64)) * 40).astype(np.float32)
    v = (rng.standard_normal((n, 1152)) * 5).astype(np.float32)
    out, vo = dev.qmf_synthesis(torch.from_numpy(X).cuda(), torch.from_numpy(v).cuda())
    out, vo = out.cpu().numpy(), vo.cpu().numpy()
    for i in range(n):
        r, rv = oracle.qmf_synthesis(X[i], v[i])
        assert np.array_equal(_bits(out[i]), _bits(r)), i
        assert np.array_equal(_bits(vo[i]), _bits(rv)), i


@pytest.mark.parametrize("n", [1, 7, 100])
def test_qmf_synthesis_downsampled_batch(pkg, oracle, dev, n):
    """sbr_qmf_synthesis with div = 1 (aacsbr.c:1175-1230), chained over two frames."""
    import torch
    rng = np.random.default_rng(150 + n)
    v = (rng.standard_normal((n, 576)) * 5).astype(np.float32)
    d_v = torch.from_numpy(v).cuda()
    for step in range(2):
        X = (rng.standard_normal((n, 2, 32, 64)) * 40).astype(np.float32)
        out, d_v = dev.qmf_synthesis_ds(torch.from_numpy(X).cuda(), d_v, scale=1.0 if step else 2.0 ** -15,
                                        bias=0.0 if step else 385.0)
        out, vo = out.cpu().numpy(), d_v.cpu().numpy()
        for i in range(n):
            r, v[i] = oracle.qmf_synthesis_ds(X[i], v[i], scale=1.0 if step else 2.0 ** -15,
                                              bias=0.0 if step else 385.0)
            assert np.array_equal(_bits(out[i]), _bits(r)), (step, i)
        assert np.array_equal(_bits(vo), _bits(v)), step


def _run_chain(pkg, oracle, dev, cfg, n, steps, seed, hdr, ps_mode="20", hdr_choice=None, fmt=None,
               check_state=True, coupling=0.0, events=None, in_place=False, downsampled=False):
    import torch
    synth = _synth()
    rng = np.random.default_rng(seed)
    fmt = pkg.PCM_F32 if fmt is None else fmt
    state = np.zeros((n, pkg.STATE_WORDS[cfg]), np.float32)
    d_state = torch.from_numpy(state).cuda()
    d_hdr = pkg.to_device(hdr)
    seen = dict(lead=0, reset=0, drop=0, ps_off=0)
    for step, fr in enumerate(synth.he_stream(rng, cfg, n, steps, hdr, ps_mode=ps_mode, hdr_choice=hdr_choice,
                                              coupling=coupling, events=events)):
        seen["lead"] += int(((fr["sbr"]["start"] == 0) & (fr["sbr"]["hdr"] == len(hdr) - 1)).sum())
        seen["drop"] += int(((fr["sbr"]["start"] == 0) & (fr["sbr"]["hdr"] != len(hdr) - 1)).sum())
        seen["reset"] += int((fr["sbr"]["reset"] == 1).sum()) if step else 0
        seen["ps_off"] += int((fr["ps"]["start"] == 0).sum()) if fr["ps"] is not None else 0
        ref_pcm, state = oracle.he_decode_batch(cfg, fr["coeffs"], fr["ics"], fr["sbr"], hdr, fr["ps"], state, fmt,
                                                downsampled=downsampled)
        pcm, d_state = dev.he_decode(cfg, torch.from_numpy(fr["coeffs"]).cuda(), pkg.to_device(fr["ics"]),
                                     pkg.to_device(fr["sbr"]), d_hdr,
                                     pkg.to_device(fr["ps"]) if fr["ps"] is not None else None,
                                     d_state, state_out=d_state if in_place else None, pcm_format=fmt,
                                     downsampled=downsampled)
        got = pcm.cpu().numpy()
        if fmt == pkg.PCM_F32:
            nbad, where = _mismatch(got, ref_pcm)
            assert nbad == 0, "step %d: %d PCM words differ, first at %s" % (step, nbad, where)
        else:
            assert np.array_equal(got, ref_pcm), "step %d int16 PCM differs" % step
        if check_state:
            nbad, where = _mismatch(d_state.cpu().numpy(), state)
            assert nbad == 0, "step %d: %d state words differ, first at %s" % (step, nbad, where)
    return seen


def test_hev1_stereo_chain_default_header(pkg, oracle, dev):
    hdr = _synth().default_headers(pkg)
    _run_chain(pkg, oracle, dev, pkg.CFG_HEV1, 40, 6, 21, hdr)


def test_hev1_stereo_chain_header_variants(pkg, oracle, dev):
    hdr = _synth().default_headers(pkg, extra=True)
    n = 63
    _run_chain(pkg, oracle, dev, pkg.CFG_HEV1, n, 6, 22, hdr, hdr_choice=np.arange(n) % len(hdr))


def test_hev1_mono_s16(pkg, oracle, dev):
    hdr = _synth().default_headers(pkg, extra=True)
    n = 21
    _run_chain(pkg, oracle, dev, pkg.CFG_HEV1_MONO, n, 4, 23, hdr, hdr_choice=np.arange(n) % len(hdr),
               fmt=pkg.PCM_S16)


def test_hev1_s16_stereo(pkg, oracle, dev):
    hdr = _synth().default_headers(pkg)
    _run_chain(pkg, oracle, dev, pkg.CFG_HEV1, 17, 3, 24, hdr, fmt=pkg.PCM_S16)


def test_hev2_chain_20band(pkg, oracle, dev):
    hdr = _synth().default_headers(pkg)
    _run_chain(pkg, oracle, dev, pkg.CFG_HEV2, 40, 6, 31, hdr, ps_mode="20")


def test_hev2_chain_34band_ipdopd(pkg, oracle, dev):
    hdr = _synth().default_headers(pkg, extra=True)
    n = 35
    _run_chain(pkg, oracle, dev, pkg.CFG_HEV2, n, 5, 32, hdr, ps_mode="34", hdr_choice=np.arange(n) % len(hdr))


def test_hev2_chain_mixed_layouts(pkg, oracle, dev):
    """10/20/34-band parameter layouts switching from frame to frame (20<->34 remap of the
    H history, delay-line resets), IPD/OPD on, mode A and mode B mixing."""
    hdr = _synth().default_headers(pkg, extra=True)
    n = 70
    _run_chain(pkg, oracle, dev, pkg.CFG_HEV2, n, 8, 33, hdr, ps_mode="mix", hdr_choice=np.arange(n) % len(hdr))


def test_hev2_s16(pkg, oracle, dev):
    hdr = _synth().default_headers(pkg)
    _run_chain(pkg, oracle, dev, pkg.CFG_HEV2, 19, 3, 34, hdr, fmt=pkg.PCM_S16)


def test_he_in_place_state(pkg, oracle, dev):
    """state_out aliasing state_in (how a decoder runs frame after frame) gives the same bits."""
    import torch
    synth = _synth()
    for cfg, ps_mode in ((pkg.CFG_HEV1, "20"), (pkg.CFG_HEV2, "mix")):
        hdr = synth.default_headers(pkg, extra=True)
        n = 28
        rng = np.random.default_rng(41)
        state = np.zeros((n, pkg.STATE_WORDS[cfg]), np.float32)
        d_state = torch.from_numpy(state).cuda()
        d_hdr = pkg.to_device(hdr)
        for fr in synth.he_stream(rng, cfg, n, 4, hdr, ps_mode=ps_mode, hdr_choice=np.arange(n) % len(hdr)):
            ref_pcm, state = oracle.he_decode_batch(cfg, fr["coeffs"], fr["ics"], fr["sbr"], hdr, fr["ps"], state)
            pcm, out = dev.he_decode(cfg, torch.from_numpy(fr["coeffs"]).cuda(), pkg.to_device(fr["ics"]),
                                     pkg.to_device(fr["sbr"]), d_hdr,
                                     pkg.to_device(fr["ps"]) if fr["ps"] is not None else None,
                                     d_state, state_out=d_state)
            assert out.data_ptr() == d_state.data_ptr()
            assert _mismatch(pcm.cpu().numpy(), ref_pcm)[0] == 0
            assert _mismatch(d_state.cpu().numpy(), state)[0] == 0


@pytest.mark.parametrize("cfgname,reps", [("CFG_HEV1", 1024), ("CFG_HEV2", 4096)])
def test_he_full_size_batch_position_independent(pkg, oracle, dev, cfgname, reps):
    """BASELINE configs 3 and 4 at full size (64 k HE-AACv1, 256 k HE-AACv2 frames): a 64-stream set,
    checked against the oracle, is tiled across the batch (several chunks of the stage workspace);
    every tile must reproduce the oracle's PCM and state bit for bit wherever it lands."""
    import torch
    synth = _synth()
    cfg = getattr(pkg, cfgname)
    base = 64
    hdr = synth.default_headers(pkg, extra=True)
    hc = np.arange(base) % len(hdr)
    rng = np.random.default_rng(77)
    frames = list(synth.he_stream(rng, cfg, base, 2, hdr, ps_mode="mix", hdr_choice=hc))
    st0 = np.zeros((base, pkg.STATE_WORDS[cfg]), np.float32)
    _, st1 = oracle.he_decode_batch(cfg, frames[0]["coeffs"], frames[0]["ics"], frames[0]["sbr"], hdr,
                                    frames[0]["ps"], st0)
    fr = frames[1]
    ref_pcm, ref_state = oracle.he_decode_batch(cfg, fr["coeffs"], fr["ics"], fr["sbr"], hdr, fr["ps"], st1,
                                                oracle.PCM_S16)
    big = pkg.Device(base * reps)
    try:
        d_state = torch.from_numpy(st1).cuda().repeat(reps, 1)
        pcm, st = big.he_decode(cfg, torch.from_numpy(fr["coeffs"]).cuda().repeat(reps, 1, 1),
                                pkg.to_device(fr["ics"]).repeat(reps), pkg.to_device(fr["sbr"]).repeat(reps),
                                pkg.to_device(hdr),
                                pkg.to_device(fr["ps"]).repeat(reps) if fr["ps"] is not None else None,
                                d_state, state_out=d_state, pcm_format=pkg.PCM_S16)
        want_pcm = torch.from_numpy(ref_pcm).cuda()
        want_st = torch.from_numpy(ref_state).cuda().view(torch.int32)
        assert bool((pcm.view(reps, base, 2048, 2) == want_pcm[None]).all())
        assert bool((st.view(torch.int32).view(reps, base, -1) == want_st[None]).all())
    finally:
        big.close()


def test_hev1_coupled_channel_pairs(pkg, oracle, dev):
    """bs_coupling = 1: sbr_dequant's pan-law branch (aacsbr.c:1094-1114), shared grid."""
    hdr = _synth().default_headers(pkg, extra=True)
    n = 42
    _run_chain(pkg, oracle, dev, pkg.CFG_HEV1, n, 5, 25, hdr, hdr_choice=np.arange(n) % len(hdr), coupling=0.7)


def test_he_decode_is_graph_capturable(pkg, oracle, dev):
    """The batched entry point neither allocates nor synchronises: a whole HE-AACv2 step can be
    captured in a HIP graph and replayed."""
    import torch
    synth = _synth()
    rng = np.random.default_rng(77)
    hdr = synth.default_headers(pkg)
    n = 48
    cfg = pkg.CFG_HEV2
    fr = next(synth.he_stream(rng, cfg, n, 1, hdr))
    state0 = np.zeros((n, pkg.STATE_WORDS[cfg]), np.float32)
    ref_pcm, ref_state = oracle.he_decode_batch(cfg, fr["coeffs"], fr["ics"], fr["sbr"], hdr, fr["ps"], state0, pkg.PCM_S16)
    args = (torch.from_numpy(fr["coeffs"]).cuda(), pkg.to_device(fr["ics"]), pkg.to_device(fr["sbr"]),
            pkg.to_device(hdr), pkg.to_device(fr["ps"]))
    st_in = torch.zeros((n, pkg.STATE_WORDS[cfg]), device="cuda")
    st_out = torch.empty_like(st_in)
    pcm = torch.empty((n, 2048, 2), dtype=torch.int16, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        dev.he_decode(cfg, *args, st_in, state_out=st_out, pcm=pcm, pcm_format=pkg.PCM_S16)   # warm-up outside capture
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            dev.he_decode(cfg, *args, st_in, state_out=st_out, pcm=pcm, pcm_format=pkg.PCM_S16)
    pcm.zero_(); st_out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(pcm.cpu().numpy(), ref_pcm)
    nbad, where = _mismatch(st_out.cpu().numpy(), ref_state)
    assert nbad == 0, where


@pytest.mark.gpu
def test_he_calls_on_two_streams_may_not_overlap(pkg):
    """One HeaacDevice = one workspace: an HE call on a second stream while the first stream's call is still in
    flight is refused (HEAAC_ERR_ARG), and accepted once that work has completed."""
    import torch
    synth = _synth()
    rng = np.random.default_rng(5)
    hdr = synth.default_headers(pkg)
    n, cfg = 32768, pkg.CFG_HEV2
    fr = next(synth.he_stream(rng, cfg, 256, 1, hdr))
    rep = n // 256
    args = (torch.from_numpy(fr["coeffs"]).cuda().repeat(rep, 1, 1), pkg.to_device(fr["ics"]).repeat(rep),
            pkg.to_device(fr["sbr"]).repeat(rep), pkg.to_device(hdr), pkg.to_device(fr["ps"]).repeat(rep))
    st = torch.zeros((n, pkg.STATE_WORDS[cfg]), device="cuda")
    pcm = torch.empty((n, 2, 2048), device="cuda")
    d = pkg.Device(n)
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(a):
        d.he_decode(cfg, *args, st, state_out=st, pcm=pcm)                 # ~1 ms of device work
    with torch.cuda.stream(b):
        with pytest.raises(pkg.HeaacError):
            d.he_decode(cfg, *args, st, state_out=st, pcm=pcm)
    a.synchronize()
    with torch.cuda.stream(b):
        d.he_decode(cfg, *args, st, state_out=st, pcm=pcm)                 # the first stream is idle now
    torch.cuda.synchronize()
    d.close()


def _random_headers(pkg, rng, count):
    """`count` valid SBR headers drawn over the whole parameter space (and three SBR rates)."""
    hs = []
    while len(hs) < count:
        try:
            hs.append(pkg.sbr_make_header(
                sample_rate=int(rng.choice([48000, 44100, 32000])),
                start_freq=int(rng.integers(0, 16)), stop_freq=int(rng.integers(0, 14)),
                xover=int(rng.integers(0, 4)), freq_scale=int(rng.integers(0, 4)),
                alter_scale=int(rng.integers(0, 2)), noise_bands=int(rng.integers(0, 4)),
                limiter_bands=int(rng.integers(0, 4)), limiter_gains=int(rng.integers(0, 4)),
                interpol_freq=int(rng.integers(0, 2)), smoothing_mode=int(rng.integers(0, 2)),
                amp_res=int(rng.integers(0, 2))))
        except ValueError:
            pass                                    # the reference rejects this combination too
    return np.concatenate(hs)


@pytest.mark.parametrize("cfg_name", ["CFG_HEV1", "CFG_HEV2"])
def test_random_sbr_headers(pkg, oracle, dev, cfg_name):
    """48 random valid headers (band layouts, patch counts, limiter / smoothing / interpolation
    modes), one stream each, four chained frames."""
    rng = np.random.default_rng(4242)
    hdr = _random_headers(pkg, rng, 48)
    n = 2 * len(hdr)
    _run_chain(pkg, oracle, dev, getattr(pkg, cfg_name), n, 4, 71, hdr, hdr_choice=np.arange(n) % len(hdr),
               ps_mode="mix")


EVENTS = dict(lead_in=3, p_switch=0.3, p_drop=0.15, p_ps_off=0.2)


@pytest.mark.parametrize("in_place", [False, True])
def test_hev1_degrade_and_transition_paths(pkg, oracle, dev, in_place):
    """Frames before the first SBR header (start = 0 on the null header: pure upsampling,
    aacsbr.c:1723-1750), mid-stream header changes (reset = 1, kx[0] / m[0] of the old header:
    sbr_x_gen's i_Temp path :1412-1446, g_temp refill :1632-1637) and unusable SBR payloads (start = 0
    mid-stream, state passes through), coupled and uncoupled channel pairs."""
    hdr = _synth().default_headers(pkg, extra=True, null=True)
    n = 72
    seen = _run_chain(pkg, oracle, dev, pkg.CFG_HEV1, n, 9, 61, hdr, hdr_choice=np.arange(n) % (len(hdr) - 1),
                      coupling=0.3, events=EVENTS, in_place=in_place)
    assert seen["lead"] > 20 and seen["reset"] > 60 and seen["drop"] > 20, seen


@pytest.mark.parametrize("ps_mode,in_place", [("20", False), ("mix", False), ("mix", True)])
def test_hev2_degrade_and_transition_paths(pkg, oracle, dev, ps_mode, in_place):
    """The same for HE-AACv2, plus PS payloads that fail to parse (ps->start = 0: L copied to R,
    aacsbr.c:1755, PS state untouched, num_env_old / is34bands_old of the last parsed frame after it)."""
    hdr = _synth().default_headers(pkg, extra=True, null=True)
    n = 72
    seen = _run_chain(pkg, oracle, dev, pkg.CFG_HEV2, n, 9, 62, hdr, ps_mode=ps_mode,
                      hdr_choice=np.arange(n) % (len(hdr) - 1), events=EVENTS, in_place=in_place)
    assert seen["lead"] > 20 and seen["reset"] > 60 and seen["drop"] > 20 and seen["ps_off"] > 60, seen


def test_hev2_degrade_paths_s16(pkg, oracle, dev):
    hdr = _synth().default_headers(pkg, extra=True, null=True)
    n = 40
    _run_chain(pkg, oracle, dev, pkg.CFG_HEV2, n, 6, 63, hdr, ps_mode="mix",
               hdr_choice=np.arange(n) % (len(hdr) - 1), events=EVENTS, fmt=pkg.PCM_S16)


@pytest.mark.parametrize("cfgname,fmtname,in_place", [("CFG_HEV1", "PCM_F32", False), ("CFG_HEV2", "PCM_S16", True),
                                                       ("CFG_HEV1_MONO", "PCM_S16", False)])
def test_downsampled_output_inside_the_decode_call(pkg, oracle, dev, cfgname, fmtname, in_place):
    """ff_sbr_apply with ext_sample_rate < sbr->sample_rate (aacsbr.c:1719): the 32-band synthesis bank
    (div = 1, :1194-1203), 1024 samples per channel, ring state in the first 576 words."""
    hdr = _synth().default_headers(pkg, extra=True)
    n = 40
    _run_chain(pkg, oracle, dev, getattr(pkg, cfgname), n, 4, 71, hdr, ps_mode="mix", hdr_choice=np.arange(n) % len(hdr),
               fmt=getattr(pkg, fmtname), in_place=in_place, downsampled=True)


def test_hev2_phase_parameters_wider_than_the_grid(pkg, oracle, dev):
    """A PS header that switches IID off leaves nr_iid_par / nr_ipdopd_par where an earlier header put them
    (aacps.c:161-172): 17 phase parameters can then meet the 20-band grid.  The reference's mixing loop runs to
    b < nr_ipdopd_par (:863) over mapped entries 11..16 that remap20 never wrote; both sides define them as 0."""
    import torch
    synth = _synth()
    rng = np.random.default_rng(91)
    hdr = synth.default_headers(pkg)
    n, cfg = 24, pkg.CFG_HEV2
    state = np.zeros((n, pkg.STATE_WORDS[cfg]), np.float32)
    d_state = torch.from_numpy(state).cuda()
    d_hdr = pkg.to_device(hdr)
    was34 = np.zeros(n, np.uint8)
    hit = 0
    for step, fr in enumerate(synth.he_stream(rng, cfg, n, 5, hdr, ps_mode="mix")):
        ps = fr["ps"]
        for s in range(n):
            if ps[s]["nr_icc_par"] != 34 and s % 3 != 2:
                ps[s]["iid_par"] = 0
                ps[s]["nr_iid_par"], ps[s]["nr_ipdopd_par"], ps[s]["enable_ipdopd"] = 34, 17, 1
                ps[s]["ipd_par"] = rng.integers(0, 8, (5, 17))
                ps[s]["opd_par"] = rng.integers(0, 8, (5, 17))
                ps[s]["is34bands"] = 0
                hit += 1
            ps[s]["is34bands_old"] = was34[s]
            was34[s] = ps[s]["is34bands"]
        assert pkg.validate_frame(cfg, fr["sbr"][:1], hdr, ps[:1]) == "NONE"
        ref_pcm, state = oracle.he_decode_batch(cfg, fr["coeffs"], fr["ics"], fr["sbr"], hdr, ps, state, pkg.PCM_F32)
        pcm, d_state = dev.he_decode(cfg, torch.from_numpy(fr["coeffs"]).cuda(), pkg.to_device(fr["ics"]),
                                     pkg.to_device(fr["sbr"]), d_hdr, pkg.to_device(ps), d_state)
        nbad, where = _mismatch(pcm.cpu().numpy(), ref_pcm)
        assert nbad == 0, "step %d: %d PCM words differ, first at %s" % (step, nbad, where)
        nbad, where = _mismatch(d_state.cpu().numpy(), state)
        assert nbad == 0, "step %d: %d state words differ, first at %s" % (step, nbad, where)
    assert hit > 20


@pytest.mark.parametrize("cfgname,ps_mode", [("CFG_HEV2", "mix"), ("CFG_HEV1", None)])
def test_long_chains_stay_bit_exact(pkg, oracle, dev, cfgname, ps_mode):
    """150 chained frames per stream with header switches, dropped payloads and PS outages along the way: every
    carried quantity (noise and sine phase indices, the smoothing history, delay lines, the PS transient detector,
    overlap and filterbank rings) goes round many times; state is handed on in place on the GPU."""
    hdr = _synth().default_headers(pkg, extra=True, null=True)
    n = 10
    seen = _run_chain(pkg, oracle, dev, getattr(pkg, cfgname), n, 150, 77, hdr, ps_mode=ps_mode or "20",
                      hdr_choice=np.arange(n) % (len(hdr) - 1), coupling=0.5,
                      events=dict(lead_in=3, p_switch=0.04, p_drop=0.03, p_ps_off=0.03), in_place=True)
    assert seen["reset"] > 20 and seen["drop"] > 10


def _header_zoo(pkg, rng, count):
    """Distinct derived headers from random header fields at every SBR rate the reference accepts."""
    seen, out = set(), []
    rates = [16000, 22050, 24000, 32000, 44100, 48000, 64000, 88200, 96000]
    tries = 0
    while len(out) < count and tries < 200000:
        tries += 1
        try:
            h = pkg.sbr_make_header(sample_rate=int(rng.choice(rates)), start_freq=int(rng.integers(0, 16)),
                                    stop_freq=int(rng.integers(0, 16)), xover=int(rng.integers(0, 8)),
                                    freq_scale=int(rng.integers(0, 4)), alter_scale=int(rng.integers(0, 2)),
                                    noise_bands=int(rng.integers(0, 4)), limiter_bands=int(rng.integers(0, 4)),
                                    limiter_gains=int(rng.integers(0, 4)), interpol_freq=int(rng.integers(0, 2)),
                                    smoothing_mode=int(rng.integers(0, 2)), amp_res=int(rng.integers(0, 2)))
        except ValueError:
            continue
        if pkg.validate_frame(pkg.CFG_HEV1, np.zeros(1, pkg.SBR_FRAME_DT), h) != "NONE":
            continue                                   # (an empty limiter table: the parser refuses these headers too)
        key = h.tobytes()
        if key not in seen:
            seen.add(key)
            out.append(h)
    return np.concatenate(out)


@pytest.mark.parametrize("cfgname", ["CFG_HEV1", "CFG_HEV2"])
def test_header_zoo(pkg, oracle, dev, cfgname):
    """300 distinct band layouts (every accepted sampling rate, crossover, patch count 1..5, 1..5 noise bands,
    limiter tables with and without the patch borders, m up to 48, kx down to the lowest the tables allow): one
    stream per header, chained, with mid-stream switches between them."""
    rng = np.random.default_rng(2025)
    hdr = _header_zoo(pkg, rng, 300)
    assert len(hdr) == 300
    assert {int(x) for x in hdr["num_patches"]} >= {1, 2, 3, 4, 5} and {int(x) for x in hdr["n_q"]} >= {1, 2, 3, 4, 5}
    assert hdr["m"].max() >= 40 and hdr["kx"].min() <= 12 and (hdr["kx"].astype(int) + hdr["m"]).max() >= 60
    n = len(hdr)
    _run_chain(pkg, oracle, dev, getattr(pkg, cfgname), n, 5, 88, hdr, ps_mode="mix", hdr_choice=np.arange(n),
               coupling=0.4, events=dict(p_switch=0.2))


@pytest.mark.parametrize("cfgname", ["CFG_HEV1", "CFG_HEV2", "CFG_LC_STEREO", "CFG_LC_MONO"])
def test_simd_float_to_int16_configuration(pkg, oracle, dev, cfgname):
    """HEAAC_PCM_S16_INTERLEAVED_SSE2: add_bias 0, sf_scale 1 / -1024 and the cvtps2dq + packssdw conversion
    (aacdec.c:577-581, x86/dsputil_mmx.c:2356-2372), spectrum 32768 x the C path's; loud frames saturate on both
    sides of the range; the downsampled bank too."""
    import torch
    synth = _synth()
    rng = np.random.default_rng(61)
    cfg = getattr(pkg, cfgname)
    n, steps = 33, 4
    big = np.float32(32768.0)
    if cfgname.startswith("CFG_LC"):
        ch = pkg.CORE_CH[cfg]
        state = np.zeros((n, ch * 512), np.float32)
        d_state = torch.from_numpy(state).cuda()
        sat = False
        for coeffs, ics in synth.lc_stream(rng, n, steps, ch):
            coeffs *= big
            coeffs[::5] *= 300.0
            ref, state = oracle.lc_decode_batch(ch, coeffs, ics, state, oracle.PCM_S16_SSE2)
            pcm, d_state = dev.lc_decode(ch, torch.from_numpy(coeffs).cuda(), pkg.to_device(ics), d_state,
                                         pcm_format=pkg.PCM_S16_SSE2)
            assert np.array_equal(pcm.cpu().numpy(), ref)
            assert np.array_equal(_bits(d_state.cpu().numpy()), _bits(state))
            sat |= bool((ref == 32767).any() and (ref == -32768).any())
        assert sat
        return
    hdr = synth.default_headers(pkg, extra=True)
    for downsampled in (False, True):
        state = np.zeros((n, pkg.STATE_WORDS[cfg]), np.float32)
        d_state = torch.from_numpy(state).cuda()
        sat = False
        for fr in synth.he_stream(rng, cfg, n, steps, hdr, ps_mode="mix", hdr_choice=np.arange(n) % len(hdr), coupling=0.5):
            coeffs = fr["coeffs"] * big
            coeffs[::5] *= 2000.0
            ref, state = oracle.he_decode_batch(cfg, coeffs, fr["ics"], fr["sbr"], hdr, fr["ps"], state,
                                                oracle.PCM_S16_SSE2, downsampled=downsampled)
            pcm, d_state = dev.he_decode(cfg, torch.from_numpy(coeffs).cuda(), pkg.to_device(fr["ics"]),
                                         pkg.to_device(fr["sbr"]), pkg.to_device(hdr),
                                         pkg.to_device(fr["ps"]) if fr["ps"] is not None else None, d_state,
                                         pcm_format=pkg.PCM_S16_SSE2, downsampled=downsampled)
            assert np.array_equal(pcm.cpu().numpy(), ref)
            assert _mismatch(d_state.cpu().numpy(), state)[0] == 0
            sat |= bool((ref == 32767).any() and (ref == -32768).any())
        assert sat


def test_unstored_x_bands_are_never_read(pkg, oracle, dev):
    """The fused HF + PS kernel leaves out the X bands it has proved to be +0 (k_psf.h: x_bands) and k_synth takes them
    from a page of zeros.  With the hand-over workspace poisoned by NaN beforehand, PCM and state still equal the
    oracle's bit for bit, over headers whose SBR range ends at band 45, 41, 38 (48 bands stored) and 27, 23 (32 bands);
    some frames do leave bands unwritten (the NaN is still there afterwards: the mechanism is exercised), and others --
    where both mixing factors of a channel can be negative -- write all of them."""
    import ctypes as C
    import torch
    synth = _synth()
    cfg = pkg.CFG_HEV2
    n = 224
    hdr = synth.default_headers(pkg, extra=True)
    hc = np.arange(n) % len(hdr)
    top16 = ((hdr["kx"].astype(int) + hdr["m"].astype(int) + 15) & ~15)[hc]
    assert set(top16.tolist()) == {32, 48}
    rng = np.random.default_rng(2024)
    frames = list(synth.he_stream(rng, cfg, n, 3, hdr, hdr_choice=hc))
    small = pkg.Device(n)
    hip = C.CDLL("libamdhip64.so")
    try:
        pW, pX, chunk = C.c_void_p(), C.c_void_p(), C.c_size_t()
        assert pkg.lib().heaac_debug_workspace(small._h, C.byref(pW), C.byref(pX), C.byref(chunk)) == 0
        assert chunk.value >= n
        xrec = 2 * 2 * 38 * 64
        state = np.zeros((n, pkg.STATE_WORDS[cfg]), np.float32)
        d_state = torch.from_numpy(state).cuda()
        d_hdr = pkg.to_device(hdr)
        skipped = {32: 0, 48: 0}
        full = 0
        for fr in frames:
            # poison X of every frame of the chunk (device to device: kind 3)
            poison = torch.full((n * xrec,), float("nan"), dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            assert hip.hipMemcpy(C.c_void_p(pX.value), C.c_void_p(poison.data_ptr()), C.c_size_t(n * xrec * 4), 3) == 0
            ref_pcm, state = oracle.he_decode_batch(cfg, fr["coeffs"], fr["ics"], fr["sbr"], hdr, fr["ps"], state)
            pcm, d_state = small.he_decode(cfg, torch.from_numpy(fr["coeffs"]).cuda(), pkg.to_device(fr["ics"]),
                                           pkg.to_device(fr["sbr"]), d_hdr, pkg.to_device(fr["ps"]), d_state)
            torch.cuda.synchronize()
            assert np.array_equal(pcm.cpu().numpy().view(np.uint32), ref_pcm.view(np.uint32))
            assert np.array_equal(d_state.cpu().numpy().view(np.uint32), state.view(np.uint32))
            back = torch.empty(n * xrec, dtype=torch.float32, device="cuda")
            assert hip.hipMemcpy(C.c_void_p(back.data_ptr()), C.c_void_p(pX.value), C.c_size_t(n * xrec * 4), 3) == 0
            x = back.cpu().numpy().reshape(n, 2, 38, 64, 2)[:, :, :32]       # [frame][channel][slot][band][re, im]
            for f in range(n):
                t = int(top16[f])
                upper = np.isnan(x[f, :, :, t:])
                assert upper.all() or not upper.any(), f              # a frame stores all of its upper bands or none
                assert not np.isnan(x[f, :, :, :t]).any(), f
                if upper.all():
                    skipped[t] += 1
                else:
                    full += 1
        assert skipped[48] > n // 2 and skipped[32] > n // 8 and full > n // 4, (skipped, full)
    finally:
        small.close()


def test_unstored_x_bands_are_never_read_hev1(pkg, oracle, dev):
    """The same for HE-AACv1, where the HF kernel itself leaves out the bands above kx + m (sbr_x_gen's literal +0) and
    says so per channel: workspace poisoned, PCM and state bit-exact, and the bands above the range (rounded up to a
    128-byte line) are still NaN afterwards in every frame whose first slots do not follow an older, wider range."""
    import ctypes as C
    import torch
    synth = _synth()
    cfg = pkg.CFG_HEV1
    n = 112
    hdr = synth.default_headers(pkg, extra=True)
    hc = np.arange(n) % len(hdr)
    top16 = ((hdr["kx"].astype(int) + hdr["m"].astype(int) + 15) & ~15)[hc]
    rng = np.random.default_rng(2025)
    frames = list(synth.he_stream(rng, cfg, n, 4, hdr, hdr_choice=hc, events=dict(p_switch=0.15)))
    small = pkg.Device(n)
    hip = C.CDLL("libamdhip64.so")
    try:
        pW, pX, chunk = C.c_void_p(), C.c_void_p(), C.c_size_t()
        assert pkg.lib().heaac_debug_workspace(small._h, C.byref(pW), C.byref(pX), C.byref(chunk)) == 0
        xrec = 2 * 38 * 64 * 2
        state = np.zeros((n, pkg.STATE_WORDS[cfg]), np.float32)
        d_state = torch.from_numpy(state).cuda()
        d_hdr = pkg.to_device(hdr)
        skipped = full = 0
        for fr in frames:
            poison = torch.full((n * xrec,), float("nan"), dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            assert hip.hipMemcpy(C.c_void_p(pX.value), C.c_void_p(poison.data_ptr()), C.c_size_t(n * xrec * 4), 3) == 0
            ref_pcm, state = oracle.he_decode_batch(cfg, fr["coeffs"], fr["ics"], fr["sbr"], hdr, None, state)
            pcm, d_state = small.he_decode(cfg, torch.from_numpy(fr["coeffs"]).cuda(), pkg.to_device(fr["ics"]),
                                           pkg.to_device(fr["sbr"]), d_hdr, None, d_state)
            torch.cuda.synchronize()
            assert np.array_equal(pcm.cpu().numpy().view(np.uint32), ref_pcm.view(np.uint32))
            assert np.array_equal(d_state.cpu().numpy().view(np.uint32), state.view(np.uint32))
            back = torch.empty(n * xrec, dtype=torch.float32, device="cuda")
            assert hip.hipMemcpy(C.c_void_p(back.data_ptr()), C.c_void_p(pX.value), C.c_size_t(n * xrec * 4), 3) == 0
            x = back.cpu().numpy().reshape(n, 2, 38, 64, 2)
            for f in range(n):
                t = int(((int(hdr[int(fr["sbr"][f]["hdr"])]["kx"]) + int(hdr[int(fr["sbr"][f]["hdr"])]["m"]) + 15) & ~15))
                for c in range(2):
                    upper = np.isnan(x[f, c, :, t:]) if t < 64 else np.zeros(1, bool)
                    assert upper.all() or not upper.any(), (f, c)
                    assert not np.isnan(x[f, c, :, :t]).any(), (f, c)
                    skipped += int(t < 64 and upper.all()); full += int(not upper.all())
        assert skipped > 4 * n and full > 0, (skipped, full)
    finally:
        small.close()
