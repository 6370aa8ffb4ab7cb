"""Debug helper (GPU box): localise the first HE stage that differs from the oracle."""
import ctypes as C, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g, oracle_lib as O
pkg = g.load_package(); synth = importlib.import_module("ffmpeg_heaac_amd.synth")
cfg = {"hev1": pkg.CFG_HEV1, "hev2": pkg.CFG_HEV2, "mono": pkg.CFG_HEV1_MONO}[sys.argv[1]]
seed = int(sys.argv[2]); n = int(sys.argv[3]); steps = int(sys.argv[4]); ps_mode = sys.argv[5] if len(sys.argv) > 5 else "20"
hdr = synth.default_headers(pkg, extra=True)
hc = np.arange(n) % len(hdr)
dev = pkg.Device()
rng = np.random.default_rng(seed)
state = np.zeros((n, pkg.STATE_WORDS[cfg]), np.float32); d_state = torch.from_numpy(state).cuda()
d_hdr = pkg.to_device(hdr)
ncore = pkg.CORE_CH[cfg]
def bits(a): return np.ascontiguousarray(a).view(np.uint32)
for step, fr in enumerate(synth.he_stream(rng, cfg, n, steps, hdr, ps_mode=ps_mode, hdr_choice=hc)):
    st_in = state.copy()
    ref_pcm, state = O.he_decode_batch(cfg, fr["coeffs"], fr["ics"], fr["sbr"], hdr, fr["ps"], state)
    pcm, d_state = dev.he_decode(cfg, torch.from_numpy(fr["coeffs"]).cuda(), pkg.to_device(fr["ics"]),
                                 pkg.to_device(fr["sbr"]), d_hdr,
                                 pkg.to_device(fr["ps"]) if fr["ps"] is not None else None, d_state)
    torch.cuda.synchronize()
    got = pcm.cpu().numpy(); gst = d_state.cpu().numpy()
    pW = C.c_void_p(); pX = C.c_void_p(); ch = C.c_size_t()
    pkg.lib().heaac_debug_workspace(dev._h, C.byref(pW), C.byref(pX), C.byref(ch))
    bad = [s for s in range(n) if not np.array_equal(bits(got[s]), bits(ref_pcm[s])) or not np.array_equal(bits(gst[s]), bits(state[s]))]
    print("step", step, "bad streams", bad[:10])
    for s in bad[:2]:
        d = O.he_decode_debug(cfg, fr["coeffs"][s], fr["ics"][s], fr["sbr"][s:s+1], hdr[hc[s]:hc[s]+1] if False else hdr, fr["ps"][s:s+1] if fr["ps"] is not None else None, st_in[s])
        # workspace copies
        Wg = np.empty((n, ncore, 32, 32, 2), np.float32); Xg = np.empty((n, 2, 38, 64, 2), np.float32)
        torch.cuda.synchronize()
        import ctypes
        tW = torch.empty(Wg.size, device="cuda"); tX = torch.empty(Xg.size, device="cuda")
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy(C.c_void_p(tW.data_ptr()), pW, C.c_size_t(Wg.size * 4), 3)
        Wg = tW.cpu().numpy().reshape(Wg.shape)
        # X layout in workspace: [frame][channel][38][64][re, im] (k_common.h, HE_X_CHANNEL); W is [unit = frame*ncore+ch]
        hip.hipMemcpy(C.c_void_p(tX.data_ptr()), pX, C.c_size_t(Xg.size * 4), 3)
        Xg = np.ascontiguousarray(np.moveaxis(tX.cpu().numpy().reshape(Xg.shape), 4, 2))     # -> [frame][channel][re/im][38][64]
        h = hdr[hc[s]]
        print(" stream", s, "hdr", hc[s], "kx", h["kx"], "m", h["m"], "n_q", h["n_q"], "interpol", h["bs_interpol_freq"], "smooth", h["bs_smoothing_mode"])
        c0 = fr["sbr"][s]["ch"][0]
        print("  num_env", c0["bs_num_env"], "t_env", c0["t_env"], "e_a", c0["e_a"], "freq_res", c0["bs_freq_res"], "harm", c0["bs_add_harmonic_flag"], "amp", c0["bs_amp_res"])
        for c in range(ncore):
            dW = bits(Wg[s, c]) != bits(d["W"][c])
            print("  ch", c, "W diff", int(dW.sum()))
        which = "X" if cfg == pkg.CFG_HEV2 else "Xsbr"
        for c in range(2):
            dX = bits(Xg[s, c][:, :32]) != bits(d[which][c][:, :32])
            idx = np.argwhere(dX)
            print("  X[%d] (vs oracle %s) diff" % (c, which), int(dX.sum()), "slots", sorted(set(idx[:, 1].tolist()))[:40], "bands", sorted(set(idx[:, 2].tolist()))[:64])
            if dX.any():
                a = idx[0]; print("   first", a, Xg[s, c][tuple(a)], d[which][c][tuple(a)])
        dp = bits(got[s]) != bits(ref_pcm[s]); print("  pcm diff", int(dp.sum()))
        ds = np.argwhere(bits(gst[s]) != bits(state[s])).ravel(); print("  state diff words", ds[:20], len(ds))
    if bad: break
