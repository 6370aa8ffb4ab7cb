"""CPU checks of the decoder's degrade and transition paths in the oracle (the GPU parity tests compare the
HIP path with it on the same streams, tests/test_he_gpu.py::test_*_degrade_*): what the path must do
follows from the reference's source and is asserted here from first principles, not from the oracle's own
stage functions alone."""
import importlib

import numpy as np


def _synth():
    import __graft_entry__ as g
    return importlib.import_module(g.PKG_NAME + ".synth")


def _lead_in(fr, hdr):
    """Every stream still in front of its first SBR header: nothing parsed, kx = 32, m = 0."""
    fr["sbr"]["hdr"] = len(hdr) - 1
    fr["sbr"]["start"] = 0; fr["sbr"]["reset"] = 0; fr["sbr"]["kx_old"] = 32; fr["sbr"]["m_old"] = 0
    fr["sbr"]["ch"] = np.zeros_like(fr["sbr"]["ch"])
    return fr


def test_pure_upsampling_leaves_the_hf_state_alone(pkg, oracle):
    """start = 0 with kx = 32, m = 0 (aacsbr.c:130, 1723-1750): no HF stage runs, so the Y tail, the gain
    history, the chirp factors and the noise / sine indices never move; the output is live audio."""
    synth = _synth()
    cfg = pkg.CFG_HEV1_MONO
    hdr = synth.default_headers(pkg, null=True)
    rng = np.random.default_rng(5)
    n, steps = 6, 3
    state = np.zeros((n, pkg.STATE_WORDS[cfg]), np.float32)
    for fr in synth.he_stream(rng, cfg, n, steps, hdr):
        pcm, state = oracle.he_decode_batch(cfg, fr["coeffs"], fr["ics"], _lead_in(fr, hdr)["sbr"], hdr, None, state)
    off_sbr = 512
    assert not state[:, off_sbr + 800:off_sbr + 1972].any()
    assert state[:, off_sbr:off_sbr + 800].any(), "analysis history and W tail are live"
    assert pcm.shape == (n, 1, 2048) and np.isfinite(pcm).all() and np.abs(pcm - 385).max() > 1e-3


def test_pure_upsampling_equals_stage_chain(pkg, oracle):
    """Same path, exact: feed the HE decoder's own analysis output (dumped W) through the 6-slot delay into
    the stage-level synthesis and demand the identical PCM."""
    synth = _synth()
    cfg = pkg.CFG_HEV1_MONO
    hdr = synth.default_headers(pkg, null=True)
    rng = np.random.default_rng(6)
    steps = 3
    state = np.zeros((1, pkg.STATE_WORDS[cfg]), np.float32)
    wtail = np.zeros((8, 32, 2), np.float32)
    v = np.zeros(1152, np.float32)
    for fr in synth.he_stream(rng, cfg, 1, steps, hdr):
        _lead_in(fr, hdr)
        dbg = oracle.he_decode_debug(cfg, fr["coeffs"][0], fr["ics"][0], fr["sbr"][0], hdr, None, state[0])
        pcm, state = oracle.he_decode_batch(cfg, fr["coeffs"], fr["ics"], fr["sbr"], hdr, None, state)
        W = dbg["W"][0]                                    # [32 slots][32 bands][2]
        hist = np.concatenate([wtail, W])
        X = np.zeros((2, 32, 64), np.float32)
        for i in range(32):
            X[0, i, :32] = hist[i + 2, :, 0]
            X[1, i, :32] = hist[i + 2, :, 1]
        out, v = oracle.qmf_synthesis(X, v)
        wtail = W[24:].copy()
        assert np.array_equal(out.view(np.uint32), pcm[0, 0].view(np.uint32))


def test_ps_off_copies_left_to_right_and_keeps_ps_state(pkg, oracle):
    """ps->start = 0 (aacsbr.c:1755): X[1] = X[0]; with equal synthesis ring states the two channels are the
    same samples, and no PS state word moves."""
    synth = _synth()
    cfg = pkg.CFG_HEV2
    hdr = synth.default_headers(pkg)
    rng = np.random.default_rng(8)
    n = 5
    state = np.zeros((n, pkg.STATE_WORDS[cfg]), np.float32)
    frames = list(synth.he_stream(rng, cfg, n, 3, hdr, ps_mode="mix"))
    for fr in frames[:2]:
        _, state = oracle.he_decode_batch(cfg, fr["coeffs"], fr["ics"], fr["sbr"], hdr, fr["ps"], state)
    off_syn = 512 + 1972
    off_ps = off_syn + 2 * 1152
    state[:, off_syn + 1152:off_syn + 2304] = state[:, off_syn:off_syn + 1152]      # same v ring in both channels
    fr = frames[2]
    fr["ps"]["start"] = 0
    pcm, st2 = oracle.he_decode_batch(cfg, fr["coeffs"], fr["ics"], fr["sbr"], hdr, fr["ps"], state)
    assert np.array_equal(pcm[:, 0].view(np.uint32), pcm[:, 1].view(np.uint32))
    assert np.array_equal(st2[:, off_ps:].view(np.uint32), state[:, off_ps:].view(np.uint32))
    assert not np.array_equal(st2[:, :off_ps], state[:, :off_ps])


def test_header_change_resets_noise_index_and_uses_old_range_for_the_carry(pkg, oracle):
    """sbr->reset (aacsbr.c:587-588, 1062-1073): f_indexnoise restarts at 0; the first i_Temp slots still
    come from the OLD header's kx / m range (sbr_x_gen :1418-1430) -- a VARFIX/FIXVAR carry into the frame
    makes i_Temp > 0, and bands above the new range are then non-zero only in those slots."""
    synth = _synth()
    cfg = pkg.CFG_HEV1_MONO
    hdr = synth.default_headers(pkg, extra=True)
    h_wide = 0                                             # kx 13, m 32
    narrow = [i for i in range(len(hdr)) if int(hdr[i]["kx"]) + int(hdr[i]["m"]) < 40]
    assert narrow, "a header with a lower top band is in the table"
    h_new = narrow[0]
    top_new = int(hdr[h_new]["kx"]) + int(hdr[h_new]["m"])
    found = False
    for seed in range(40):
        rng = np.random.default_rng(100 + seed)
        state = np.zeros((1, pkg.STATE_WORDS[cfg]), np.float32)
        frames = list(synth.he_stream(rng, cfg, 1, 3, hdr, hdr_choice=[h_wide]))
        for fr in frames[:2]:
            _, state = oracle.he_decode_batch(cfg, fr["coeffs"], fr["ics"], fr["sbr"], hdr, None, state)
        fr = frames[2]
        t_old = int(fr["sbr"]["ch"][0, 0]["t_env_num_env_old"])
        if t_old <= 16:
            continue                                       # no carry: i_Temp = 0
        # switch the header for this frame: regenerate its grid under the new header's amp_res rule is not
        # needed (the grid is header independent); only the header index, reset and the old range change
        fr["sbr"]["hdr"] = h_new
        fr["sbr"]["reset"] = 1
        fr["sbr"]["kx_old"], fr["sbr"]["m_old"] = hdr[h_wide]["kx"], hdr[h_wide]["m"]
        dbg = oracle.he_decode_debug(cfg, fr["coeffs"][0], fr["ics"][0], fr["sbr"][0], hdr, None, state[0])
        _, st2 = oracle.he_decode_batch(cfg, fr["coeffs"], fr["ics"], fr["sbr"], hdr, None, state)
        X = dbg["X"][0]                                    # [re/im][38][64]
        i_temp = 2 * t_old - 32
        above = X[:, :, top_new:45]
        assert np.abs(above[:, :i_temp]).max() > 0, "the carry slots still hold the old, wider range"
        assert not above[:, i_temp:32].any(), "the new range ends at kx + m of the new header"
        slots = 2 * (int(fr["sbr"]["ch"][0, 0]["t_env"][int(fr["sbr"]["ch"][0, 0]["bs_num_env"])]) -
                     int(fr["sbr"]["ch"][0, 0]["t_env"][0]))
        idx = st2[0, 512 + 1957:512 + 1958].view(np.uint32)[0]
        assert idx == (slots * int(hdr[h_new]["m"])) & 0x1ff, "f_indexnoise counted from 0 after the reset"
        found = True
        break
    assert found, "no seed produced a carry into the third frame"
