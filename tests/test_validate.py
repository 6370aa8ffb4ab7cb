"""Record validation at the C-ABI boundary (include/heaac_dsp.h, csrc/validate.h).  CPU: every synthetic
frame the generators emit is valid and every single-field poisoning is named; GPU: a poisoned record in a
batch is reported with its index, and decoding it anyway (header index past the table) does not fault."""
import importlib

import numpy as np
import pytest


def _synth():
    import __graft_entry__ as g
    return importlib.import_module(g.PKG_NAME + ".synth")


def _frames(pkg, cfg, n=24, steps=4, seed=3, events=None, ps_mode="mix"):
    synth = _synth()
    hdr = synth.default_headers(pkg, extra=True, null=True)
    rng = np.random.default_rng(seed)
    return hdr, list(synth.he_stream(rng, cfg, n, steps, hdr, ps_mode=ps_mode,
                                     hdr_choice=np.arange(n) % (len(hdr) - 1), coupling=0.3, events=events))


@pytest.mark.parametrize("cfgname", ["CFG_HEV1", "CFG_HEV1_MONO", "CFG_HEV2"])
def test_generated_streams_are_valid(pkg, cfgname):
    cfg = getattr(pkg, cfgname)
    ev = dict(lead_in=2, p_switch=0.3, p_drop=0.2, p_ps_off=0.2)
    hdr, frames = _frames(pkg, cfg, events=ev)
    for fr in frames:
        for s in range(len(fr["sbr"])):
            ps = fr["ps"][s] if fr["ps"] is not None else None
            assert pkg.validate_frame(cfg, fr["sbr"][s], hdr, ps) == "NONE", (s, fr["sbr"][s]["hdr"])


def _good_hev2(pkg):
    cfg = pkg.CFG_HEV2
    hdr, frames = _frames(pkg, cfg, n=8, steps=2, ps_mode="20")
    fr = frames[1]
    for s in range(8):
        if fr["sbr"][s]["ch"][0]["bs_num_env"] >= 2:
            return cfg, hdr, fr["sbr"][s].copy(), fr["ps"][s].copy()
    raise AssertionError("no multi-envelope frame")


POISON_SBR = [
    ("HDR_INDEX", lambda f, h: f.__setitem__("hdr", len(h))),
    ("SBR_NUM_ENV", lambda f, h: f["ch"][0].__setitem__("bs_num_env", 6)),
    ("SBR_NUM_ENV", lambda f, h: f["ch"][0].__setitem__("bs_num_env", 0)),
    ("SBR_NUM_ENV", lambda f, h: f["ch"][0].__setitem__("bs_num_noise", 3)),
    ("SBR_T_ENV", lambda f, h: f["ch"][0]["t_env"].__setitem__(1, 0)),                       # not increasing
    ("SBR_T_ENV", lambda f, h: f["ch"][0]["t_env"].__setitem__(int(f["ch"][0]["bs_num_env"]), 20)),
    ("SBR_T_ENV", lambda f, h: f["ch"][0]["t_env"].__setitem__(0, 200)),
    ("SBR_T_Q", lambda f, h: f["ch"][0]["t_q"].__setitem__(1, 20)),
    ("SBR_FLAGS", lambda f, h: f["ch"][0]["bs_invf_mode"][0].__setitem__(0, 4)),
    ("SBR_FLAGS", lambda f, h: f["ch"][0]["e_a"].__setitem__(1, 9)),
    ("SBR_FLAGS", lambda f, h: f.__setitem__("bs_coupling", 1)),                           # coupling on an SCE
    ("SBR_OLD_RANGE", lambda f, h: f.__setitem__("kx_old", 40)),
    ("SBR_OLD_RANGE", lambda f, h: f["ch"][0].__setitem__("t_env_num_env_old", 30)),
]
POISON_PS = [
    ("PS_NUM_ENV", lambda p: p.__setitem__("num_env", 6)),
    ("PS_NUM_ENV", lambda p: p.__setitem__("num_env", 0)),
    ("PS_BORDER", lambda p: p["border_position"].__setitem__(0, 0)),
    ("PS_BORDER", lambda p: p["border_position"].__setitem__(int(p["num_env"]), 30)),
    ("PS_NR_PAR", lambda p: p.__setitem__("nr_iid_par", 21)),
    ("PS_NR_PAR", lambda p: p.__setitem__("icc_mode", 6)),
    ("PS_PAR", lambda p: p["iid_par"][0].__setitem__(3, 8)),
    ("PS_PAR", lambda p: p["icc_par"][0].__setitem__(0, -1)),
]
POISON_HDR = [
    ("HDR_RANGE", lambda h: h.__setitem__("kx", 33)),
    ("HDR_RANGE", lambda h: h.__setitem__("m", 60)),
    ("HDR_COUNTS", lambda h: h.__setitem__("n_q", 6)),
    ("HDR_COUNTS", lambda h: h.__setitem__("n_lim", 30)),
    ("HDR_TABLE", lambda h: h["f_tablehigh"].__setitem__(2, int(h["f_tablehigh"][1]))),
    ("HDR_TABLE", lambda h: h["f_tablelim"].__setitem__(0, 0)),
    ("HDR_MAP", lambda h: h["map_src"].__setitem__(int(h["kx"]) + 1, 40)),
    ("HDR_MAP", lambda h: h["map_hi"].__setitem__(int(h["kx"]), 0xff)),
    ("HDR_FLAGS", lambda h: h.__setitem__("bs_limiter_gains", 4)),
]


def test_every_poisoned_field_is_named(pkg):
    cfg, hdr, sbr, ps = _good_hev2(pkg)
    assert pkg.validate_frame(cfg, sbr, hdr, ps) == "NONE"
    for want, poison in POISON_SBR:
        f = sbr.copy(); poison(f, hdr)
        assert pkg.validate_frame(cfg, f, hdr, ps) == want, want
    for want, poison in POISON_PS:
        p = ps.copy(); poison(p)
        assert pkg.validate_frame(cfg, sbr, hdr, p) == want, want
    for want, poison in POISON_HDR:
        h = hdr.copy(); poison(h[int(sbr["hdr"])])
        assert pkg.validate_frame(cfg, sbr, h, ps) == want, want
    # start = 1 on the header-less state
    f = sbr.copy(); f["hdr"] = len(hdr) - 1
    assert pkg.validate_frame(cfg, f, hdr, ps) == "HDR_UNSTARTED"
    # ... which is fine for a frame in front of the first header, whatever its channel records hold
    f["start"] = 0; f["kx_old"] = 32; f["m_old"] = 0; f["ch"][0]["bs_num_env"] = 77
    assert pkg.validate_frame(cfg, f, hdr, ps) == "NONE"
    # PS off: nothing of the record is read
    p = ps.copy(); p["start"] = 0; p["num_env"] = 9
    assert pkg.validate_frame(cfg, sbr, hdr, p) == "NONE"


def test_make_header_outputs_are_valid_over_the_parameter_space(pkg):
    """Every header heaac_sbr_make_header accepts passes the checker (and so can be uploaded)."""
    synth = _synth()
    cfg, hdr, sbr, ps = _good_hev2(pkg)
    sbr["hdr"] = 0; sbr["start"] = 0
    ok = 0
    for start in range(0, 16, 3):
        for stop in range(0, 14, 2):
            for xover in (0, 2, 5):
                for fs in range(4):
                    try:
                        h = pkg.sbr_make_header(start_freq=start, stop_freq=stop, xover=xover, freq_scale=fs,
                                                alter_scale=fs & 1, noise_bands=fs, limiter_bands=(fs + 1) & 3)
                    except Exception:
                        continue
                    ok += 1
                    assert pkg.validate_frame(cfg, sbr, h, ps) == "NONE", (start, stop, xover, fs)
    assert ok > 50


@pytest.mark.gpu
def test_poisoned_record_in_a_batch_is_reported_not_faulted(pkg, dev):
    import torch
    cfg = pkg.CFG_HEV2
    hdr, frames = _frames(pkg, cfg, n=300, steps=2, ps_mode="20")
    fr = frames[1]
    d_hdr = pkg.to_device(hdr)
    assert dev.he_check(cfg, pkg.to_device(fr["sbr"]), d_hdr, pkg.to_device(fr["ps"])) is None
    bad = fr["sbr"].copy()
    k = int(np.flatnonzero(bad["start"] == 1)[5])
    bad[k]["ch"][0]["t_env"][1] = 99
    bad[k + 7]["hdr"] = 60000
    assert dev.he_check(cfg, pkg.to_device(bad), d_hdr, pkg.to_device(fr["ps"])) == (k, "SBR_T_ENV")
    badps = fr["ps"].copy(); badps[11]["num_env"] = 8
    assert dev.he_check(cfg, pkg.to_device(fr["sbr"]), d_hdr, pkg.to_device(badps)) == (11, "PS_NUM_ENV")
    # decoding a batch whose only defect is a header index past the table must not fault: the index is
    # clamped, every OTHER frame decodes exactly as in the clean batch
    only_idx = fr["sbr"].copy(); only_idx[k + 7]["hdr"] = 60000
    st = torch.zeros((300, pkg.STATE_WORDS[cfg]), device="cuda")
    args = (torch.from_numpy(fr["coeffs"]).cuda(), pkg.to_device(fr["ics"]))
    pcm0, _ = dev.he_decode(cfg, *args, pkg.to_device(fr["sbr"]), d_hdr, pkg.to_device(fr["ps"]), st)
    pcm1, _ = dev.he_decode(cfg, *args, pkg.to_device(only_idx), d_hdr, pkg.to_device(fr["ps"]), st)
    torch.cuda.synchronize()
    keep = torch.ones(300, dtype=torch.bool, device="cuda"); keep[k + 7] = False
    assert bool((pcm0[keep].view(torch.int32) == pcm1[keep].view(torch.int32)).all())
