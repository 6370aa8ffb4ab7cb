"""GPU parity: batched IMDCT and AAC-LC synthesis (through the C ABI) vs the oracle.
Bar: bit-exact floats, identical int16."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _synth(pkg):
    import importlib
    return importlib.import_module("ffmpeg_heaac_amd.synth")


@pytest.mark.parametrize("which,half", [(0, 1024), (1, 128), (2, 64), (3, 64)])
@pytest.mark.parametrize("n", [1, 7, 8, 257])
def test_imdct_half_batch(pkg, oracle, dev, which, half, n):
    import torch
    rng = np.random.default_rng(100 + which * 10 + n)
    x = (rng.standard_normal((n, half)) * rng.choice([1e-6, 1.0, 300.0], (n, 1))).astype(np.float32)
    x[0, :4] = [0.0, -0.0, 1e-40, -1e-42]          # zeros and denormals
    ref = oracle.imdct_half(which, x)
    out = dev.imdct_half(which, torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.array_equal(_bits(out), _bits(ref))


def test_imdct_half_rejects_alias(pkg, dev):
    import torch
    x = torch.zeros((4, 1024), device="cuda")
    with pytest.raises(pkg.HeaacError):
        import ctypes as C
        rc = pkg.lib().heaac_imdct_half_batch(dev._h, 0, C.c_void_p(x.data_ptr()), C.c_void_p(x.data_ptr()),
                                             C.c_size_t(4), None)
        pkg._check(rc, "alias")


@pytest.mark.parametrize("channels", [1, 2])
@pytest.mark.parametrize("fmt", ["f32", "s16"])
def test_lc_chained_streams(pkg, oracle, dev, channels, fmt):
    """6 chained frames x 96 streams covering every window-sequence transition."""
    import torch
    synth = _synth(pkg)
    rng = np.random.default_rng(11 + channels)
    n, steps = 96, 6
    pf = pkg.PCM_F32 if fmt == "f32" else pkg.PCM_S16
    state = (rng.standard_normal((n, channels * 512)) * 1e-3).astype(np.float32)
    d_state = torch.from_numpy(state).cuda()
    seen = set()
    for coeffs, ics in synth.lc_stream(rng, n, steps, channels):
        # loud frames now and then so the int16 saturation branch runs
        coeffs[::17] *= 200.0
        seen |= set(map(tuple, ics["window_sequence"].reshape(-1, 2)))
        ref_pcm, state = oracle.lc_decode_batch(channels, coeffs, ics, state, pf)
        pcm, d_state = dev.lc_decode(channels, torch.from_numpy(coeffs).cuda(), pkg.to_device(ics), d_state,
                                     pcm_format=pf)
        got = pcm.cpu().numpy()
        if fmt == "f32":
            assert np.array_equal(_bits(got), _bits(ref_pcm))
        else:
            assert np.array_equal(got, ref_pcm)
            assert (np.abs(ref_pcm.astype(int)) == 32767).any() or (ref_pcm == -32768).any()
        assert np.array_equal(_bits(d_state.cpu().numpy()), _bits(state))
    # ONLY_LONG->ONLY_LONG, ->START, START->SHORT, SHORT->SHORT/STOP, STOP->LONG all exercised
    assert {(0, 0), (1, 0), (2, 1), (3, 2), (0, 3)} <= seen


def test_lc_in_place_state_and_odd_sizes(pkg, oracle, dev):
    import torch
    synth = _synth(pkg)
    for n in (1, 3, 5000):
        rng = np.random.default_rng(n)
        coeffs, ics = next(synth.lc_stream(rng, n, 1, 2))
        state = (rng.standard_normal((n, 1024)) * 1e-3).astype(np.float32)
        ref_pcm, ref_state = oracle.lc_decode_batch(2, coeffs, ics, state, pkg.PCM_F32)
        d_state = torch.from_numpy(state).cuda()
        pcm, out_state = dev.lc_decode(2, torch.from_numpy(coeffs).cuda(), pkg.to_device(ics), d_state,
                                       state_out=d_state)
        assert out_state.data_ptr() == d_state.data_ptr()
        assert np.array_equal(_bits(pcm.cpu().numpy()), _bits(ref_pcm))
        assert np.array_equal(_bits(d_state.cpu().numpy()), _bits(ref_state))


@pytest.mark.parametrize("fmt", ["f32", "s16"])
def test_lc_mono_odd_counts(pkg, oracle, dev, fmt):
    """SCE frames are paired two per wavefront: odd counts leave the last one alone."""
    import torch
    synth = _synth(pkg)
    pf = pkg.PCM_F32 if fmt == "f32" else pkg.PCM_S16
    for n in (1, 3, 77):
        rng = np.random.default_rng(100 + n)
        state = (rng.standard_normal((n, 512)) * 1e-3).astype(np.float32)
        d_state = torch.from_numpy(state).cuda()
        for coeffs, ics in synth.lc_stream(rng, n, 3, 1):
            ref_pcm, state = oracle.lc_decode_batch(1, coeffs, ics, state, pf)
            pcm, d_state = dev.lc_decode(1, torch.from_numpy(coeffs).cuda(), pkg.to_device(ics), d_state,
                                         pcm_format=pf)
            got = pcm.cpu().numpy()
            assert np.array_equal(_bits(got), _bits(ref_pcm)) if fmt == "f32" else np.array_equal(got, ref_pcm)
            assert np.array_equal(_bits(d_state.cpu().numpy()), _bits(state))


def test_lc_empty_batch(pkg, dev):
    import torch
    z = torch.zeros((0, 2, 1024), device="cuda")
    pcm, st = dev.lc_decode(2, z, torch.zeros(0, dtype=torch.uint8, device="cuda"),
                            torch.zeros((0, 1024), device="cuda"))
    assert pcm.shape[0] == 0


def test_lc_full_size_batch_position_independent(pkg, oracle, dev):
    """BASELINE config 2 size (64 k stereo frames): a 64-frame oracle-checked set is
    tiled across the batch; every tile must reproduce the oracle bit for bit, wherever
    in the grid it lands (units are independent)."""
    import torch
    synth = _synth(pkg)
    rng = np.random.default_rng(2)
    base, reps = 64, 1024
    coeffs, ics = next(synth.lc_stream(rng, base, 1, 2))
    ics["window_sequence"][:, :, 0] = rng.integers(0, 4, (base, 2))
    ics["window_sequence"][:, :, 1] = rng.integers(0, 4, (base, 2))
    state = (rng.standard_normal((base, 1024)) * 1e-3).astype(np.float32)
    ref_pcm, ref_state = oracle.lc_decode_batch(2, coeffs, ics, state, pkg.PCM_S16)
    d_coeffs = torch.from_numpy(coeffs).cuda().repeat(reps, 1, 1)
    d_ics = pkg.to_device(ics).repeat(reps)
    d_state = torch.from_numpy(state).cuda().repeat(reps, 1)
    pcm, st = dev.lc_decode(2, d_coeffs, d_ics, d_state, pcm_format=pkg.PCM_S16)
    pcm = pcm.view(reps, base, 1024, 2)
    st = st.view(reps, base, 1024)
    want_pcm = torch.from_numpy(ref_pcm).cuda()
    want_st = torch.from_numpy(ref_state).cuda()
    assert bool((pcm == want_pcm[None]).all())
    assert bool((st.view(torch.int32) == want_st.view(torch.int32)[None]).all())


@pytest.mark.parametrize("channels", [1, 2])
def test_independent_coupling_after_the_imdct(pkg, oracle, dev, channels):
    """spectral_to_sample with a coupling element at AFTER_IMDCT (aacdec.c:1907-1931): the coupling channel's own
    IMDCT, the target element's IMDCT, dest += gain * (src - bias), float_to_int16_interleave -- chained over
    frames, some targets not coupled, two coupling elements on one target applied in order."""
    import torch
    synth = _synth(pkg)
    rng = np.random.default_rng(41 + channels)
    n, steps = 50, 4
    state = np.zeros((n, channels * 512), np.float32)
    cstate = np.zeros((2, n, 512), np.float32)
    d_state, d_cstate = torch.from_numpy(state).cuda(), torch.from_numpy(cstate).cuda()
    tgt = synth.lc_stream(rng, n, steps, channels)
    cc = [synth.lc_stream(rng, n, steps, 1), synth.lc_stream(rng, n, steps, 1)]
    for step in range(steps):
        coeffs, ics = next(tgt)
        coeffs[::7] *= 1000.0                                 # loud: the int16 clip runs on coupled sums too
        ref, state = oracle.lc_decode_batch(channels, coeffs, ics, state, pkg.PCM_F32)
        pcm, d_state = dev.lc_decode(channels, torch.from_numpy(coeffs).cuda(), pkg.to_device(ics), d_state,
                                     pcm_format=pkg.PCM_F32)
        ref16 = got16 = None
        for e in range(2):
            ccoef, cics = next(cc[e])
            cref, cstate[e] = oracle.lc_decode_batch(1, ccoef, cics, cstate[e], pkg.PCM_F32)
            cpcm, cs = dev.lc_decode(1, torch.from_numpy(ccoef).cuda(), pkg.to_device(cics), d_cstate[e],
                                     pcm_format=pkg.PCM_F32)
            d_cstate[e] = cs
            cpl = np.zeros(n, pkg.COUPLING_DT)
            cpl["gain"] = (2.0 ** (rng.integers(-12, 5, (n, 2)) / 4.0)).astype(np.float32)    # pow(scale, -gain)
            cpl["on"] = rng.random((n, 2)) < 0.7
            cpl["on"][0] = 0
            cpl["gain"][0] = np.nan                            # an uncoupled channel's gain is never read
            ref, ref16 = oracle.couple_after_imdct_batch(channels, ref, cref.reshape(n, 1024), cpl, s16=e == 1)
            got16 = dev.couple_after_imdct(channels, pcm, cpcm, pkg.to_device(cpl), s16=e == 1)
        assert np.array_equal(_bits(pcm.cpu().numpy()), _bits(ref)), step
        assert np.array_equal(got16.cpu().numpy(), ref16), step
        assert (np.abs(ref16.astype(int)) == 32767).any() or (ref16 == -32768).any()
        assert np.array_equal(_bits(d_state.cpu().numpy()), _bits(state))
