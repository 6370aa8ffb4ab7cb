"""Spectral tools before the IMDCT: HIP path vs oracle, bit-exact."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _synth():
    import __graft_entry__ as g
    return importlib.import_module(g.PKG_NAME + ".synth")


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("channels", [1, 2])
@pytest.mark.parametrize("n", [1, 7, 300])
def test_spectral_tools_match_oracle(pkg, oracle, dev, channels, n):
    import torch
    rng = np.random.default_rng(7 * n + channels)
    tools = _synth().tools_frames(rng, pkg, n, channels)
    coeffs = (rng.standard_normal((n, channels, 1024)) * 1e-4).astype(np.float32)
    ref = oracle.spectral_tools_batch(channels, coeffs, tools)
    d = torch.from_numpy(coeffs).cuda()
    dev.spectral_tools(channels, d, pkg.to_device(tools))
    assert np.array_equal(_bits(d.cpu().numpy()), _bits(ref))
    # the tools did something on this batch
    assert n < 7 or not np.array_equal(_bits(ref), _bits(coeffs))


@pytest.mark.parametrize("channels", [1, 2])
def test_spectral_tools_with_noise_substitution(pkg, oracle, dev, channels):
    """PNS first (generator state in / out per stream), chained over three frames."""
    import torch
    n = 150
    rng = np.random.default_rng(40 + channels)
    state = rng.integers(-2**31, 2**31, n).astype(np.int32)
    state[0] = 0x1f2e3d4c                                   # ac->random_state at init
    d_state = torch.from_numpy(state.copy()).cuda()
    for step in range(3):
        tools = _synth().tools_frames(rng, pkg, n, channels)
        coeffs = (rng.standard_normal((n, channels, 1024)) * 1e-4).astype(np.float32)
        ref, state = oracle.spectral_tools_batch(channels, coeffs, tools, state)
        d = torch.from_numpy(coeffs).cuda()
        dev.spectral_tools(channels, d, pkg.to_device(tools), rng=d_state)
        assert np.array_equal(_bits(d.cpu().numpy()), _bits(ref)), "step %d" % step
        assert np.array_equal(d_state.cpu().numpy(), state), "step %d" % step


@pytest.mark.parametrize("channels", [1, 2])
def test_spectral_tools_with_main_prediction(pkg, oracle, dev, channels):
    """AAC-Main backward-adaptive prediction (with PNS ahead of it), predictor state chained
    over five frames from reset_all_predictors."""
    import torch
    n = 60
    rng = np.random.default_rng(60 + channels)
    rs = np.full(n, 0x1f2e3d4c, np.int32)
    pred = np.zeros((n, channels, pkg.MAX_PREDICTORS), pkg.PRED_STATE_DT)
    pred["var0"] = 1.0; pred["var1"] = 1.0
    pred = pred.view(np.float32).reshape(n, channels, pkg.MAX_PREDICTORS, 6)
    d_rs = torch.from_numpy(rs.copy()).cuda()
    d_pred = torch.from_numpy(pred.copy()).cuda()
    active = False
    for step in range(5):
        tools = _synth().tools_frames(rng, pkg, n, channels)
        coeffs = (rng.standard_normal((n, channels, 1024)) * 1e-4).astype(np.float32)
        ref, rs, pred = oracle.spectral_tools_batch(channels, coeffs, tools, rs, pred)
        d = torch.from_numpy(coeffs).cuda()
        dev.spectral_tools(channels, d, pkg.to_device(tools), rng=d_rs, pred=d_pred)
        assert np.array_equal(_bits(d.cpu().numpy()), _bits(ref)), "step %d" % step
        assert np.array_equal(d_rs.cpu().numpy(), rs), "step %d" % step
        assert np.array_equal(_bits(d_pred.cpu().numpy()), _bits(pred)), "step %d" % step
        active = active or bool((pred[..., 2] > 1).any())          # var0 > 1: predictors have adapted
    assert active


def test_spectral_tools_then_lc_decode(pkg, oracle, dev):
    """tools -> imdct_and_windowing: the prefix of spectral_to_sample (aacdec.c:1903-1925)."""
    import torch
    synth = _synth()
    rng = np.random.default_rng(99)
    n = 64
    coeffs, ics = next(synth.lc_stream(rng, n, 1, 2))
    tools = synth.tools_frames(rng, pkg, n, 2)
    # keep the window layout of the tools consistent with the frame's window sequence
    state = (rng.standard_normal((n, 1024)) * 1e-3).astype(np.float32)
    ref_c = oracle.spectral_tools_batch(2, coeffs, tools)
    ref_pcm, ref_state = oracle.lc_decode_batch(2, ref_c, ics, state, pkg.PCM_S16)
    d = torch.from_numpy(coeffs).cuda()
    dev.spectral_tools(2, d, pkg.to_device(tools))
    pcm, st = dev.lc_decode(2, d, pkg.to_device(ics), torch.from_numpy(state).cuda(), pcm_format=pkg.PCM_S16)
    assert np.array_equal(pcm.cpu().numpy(), ref_pcm)
    assert np.array_equal(_bits(st.cpu().numpy()), _bits(ref_state))


def test_spectral_tools_empty_and_bad_args(pkg, dev):
    import torch
    z = torch.zeros((0, 2, 1024), device="cuda")
    dev.spectral_tools(2, z, torch.zeros(0, dtype=torch.uint8, device="cuda"))
    with pytest.raises(pkg.HeaacError):
        pkg._check(pkg.lib().heaac_spectral_tools_batch(dev._h, 3, None, None, None, None, None, None, 1, None),
                   "bad channels")
