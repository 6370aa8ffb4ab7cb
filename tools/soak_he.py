"""Randomised parity soak of the HE paths on the GPU against the oracle (run through gpurun; prints one line per
configuration and exits non-zero at the first mismatch).  Wider than the test suite's chains: many seeds, all header
variants, header changes / dropped payloads / PS-off events, baseline and mixed PS layouts, in place and out of place,
with the X hand-over workspace poisoned by NaN before every step (unstored +0 bands must never be read).
    python3 tools/soak_he.py [seconds]"""
import ctypes as C, importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
pkg = g.load_package(); synth = importlib.import_module("ffmpeg_heaac_amd.synth")
import oracle_lib as oracle
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
hip = C.CDLL("libamdhip64.so")
n = 96
dev = pkg.Device(n)
pW, pX, chunk = C.c_void_p(), C.c_void_p(), C.c_size_t()
assert pkg.lib().heaac_debug_workspace(dev._h, C.byref(pW), C.byref(pX), C.byref(chunk)) == 0
xfloats = n * 2 * 2 * 38 * 64
poison = torch.full((xfloats,), float("nan"), dtype=torch.float32, device="cuda")
hdr = synth.default_headers(pkg, extra=True)
d_hdr = pkg.to_device(hdr)
t0 = time.time(); frames = 0; seed = 5000; configs = 0
while time.time() - t0 < budget:
    seed += 1
    rng = np.random.default_rng(seed)
    cfg = [pkg.CFG_HEV2, pkg.CFG_HEV2, pkg.CFG_HEV1, pkg.CFG_HEV1_MONO][seed % 4]
    ps_mode = "20" if seed % 8 < 5 else "mix"
    in_place = bool(seed & 16)
    events = dict(p_switch=0.1, p_drop=0.05, p_ps_off=0.05) if seed % 3 == 0 else None
    hc = rng.integers(0, len(hdr), n)
    state = np.zeros((n, pkg.STATE_WORDS[cfg]), np.float32)
    d_state = torch.from_numpy(state).cuda()
    kw = dict(ps_mode=ps_mode, hdr_choice=hc, coupling=0.3 if cfg == pkg.CFG_HEV1 else 0.0)
    if events:
        kw["events"] = events
    stream = synth.he_stream(rng, cfg, n, 8, hdr, **kw)
    for step, fr in enumerate(stream):
        torch.cuda.synchronize()
        assert hip.hipMemcpy(C.c_void_p(pX.value), C.c_void_p(poison.data_ptr()), C.c_size_t(xfloats * 4), 3) == 0
        ref_pcm, state = oracle.he_decode_batch(cfg, fr["coeffs"], fr["ics"], fr["sbr"], hdr, fr["ps"], state)
        pcm, d_state = dev.he_decode(cfg, torch.from_numpy(fr["coeffs"]).cuda(), pkg.to_device(fr["ics"]),
                                     pkg.to_device(fr["sbr"]), d_hdr,
                                     pkg.to_device(fr["ps"]) if fr["ps"] is not None else None,
                                     d_state, state_out=d_state if in_place else None)
        got, gst = pcm.cpu().numpy(), d_state.cpu().numpy()
        ok = np.array_equal(got.view(np.uint32), ref_pcm.view(np.uint32)) and np.array_equal(gst.view(np.uint32), state.view(np.uint32))
        if not ok:
            bad = np.argwhere(got.view(np.uint32) != ref_pcm.view(np.uint32))
            print("MISMATCH seed %d cfg %d ps %s step %d: %d PCM words, first %s" % (seed, cfg, ps_mode, step, len(bad), bad[:3].tolist()))
            sys.exit(1)
        frames += n
    configs += 1
print("soak ok: %d configurations, %d frames, %.0f s" % (configs, frames, time.time() - t0))
