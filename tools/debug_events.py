"""Find the first frame of an event stream (synth.he_stream events) where the HIP path and the oracle
part, and print what the frame was.  CFG=hev1|hev2 SEED=.. PS=mix python3 tools/debug_events.py"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
import oracle_lib as O
pkg = g.load_package(); synth = importlib.import_module("ffmpeg_heaac_amd.synth")
cfg = pkg.CFG_HEV1 if os.environ.get("CFG", "hev2") == "hev1" else pkg.CFG_HEV2
ev = dict(lead_in=int(os.environ.get("LEAD", 3)), p_switch=float(os.environ.get("SW", 0.3)),
          p_drop=float(os.environ.get("DROP", 0.15)), p_ps_off=float(os.environ.get("PSOFF", 0.2)))
hdr = synth.default_headers(pkg, extra=True, null=True)
n, steps = 72, 9
rng = np.random.default_rng(int(os.environ.get("SEED", 62)))
dev = pkg.Device()
state = np.zeros((n, pkg.STATE_WORDS[cfg]), np.float32)
d_state = torch.from_numpy(state).cuda(); d_hdr = pkg.to_device(hdr)
bad_streams = set()
for step, fr in enumerate(synth.he_stream(rng, cfg, n, steps, hdr, ps_mode=os.environ.get("PS", "mix"),
                                          hdr_choice=np.arange(n) % (len(hdr) - 1), coupling=0.3, events=ev)):
    ref_pcm, state = O.he_decode_batch(cfg, fr["coeffs"], fr["ics"], fr["sbr"], hdr, fr["ps"], state, O.PCM_F32)
    pcm, d_state = dev.he_decode(cfg, torch.from_numpy(fr["coeffs"]).cuda(), pkg.to_device(fr["ics"]),
                                 pkg.to_device(fr["sbr"]), d_hdr,
                                 pkg.to_device(fr["ps"]) if fr["ps"] is not None else None, d_state)
    got = pcm.cpu().numpy(); gst = d_state.cpu().numpy()
    for s in range(n):
        if s in bad_streams:
            continue
        pb = (got[s].view(np.uint32) != ref_pcm[s].view(np.uint32))
        sb = (gst[s].view(np.uint32) != state[s].view(np.uint32))
        if pb.any() or sb.any():
            bad_streams.add(s)
            f = fr["sbr"][s]; h = hdr[int(f["hdr"])]
            print("step %d stream %d: pcm bad %d (ch %s, first idx %s) state bad %d (first words %s)" % (
                step, s, pb.sum(), np.unique(np.argwhere(pb)[:, 0]).tolist() if pb.any() else [],
                np.argwhere(pb)[:3].tolist(), sb.sum(), np.flatnonzero(sb)[:6].tolist()))
            print("   sbr: hdr %d start %d reset %d kx_old %d m_old %d coupling %d | hdr kx %d m %d smoothing %d | ch0 t_old %d num_env %d t_env %s e_a %s" % (
                f["hdr"], f["start"], f["reset"], f["kx_old"], f["m_old"], f["bs_coupling"], h["kx"], h["m"],
                h["bs_smoothing_mode"], f["ch"][0]["t_env_num_env_old"], f["ch"][0]["bs_num_env"],
                f["ch"][0]["t_env"][:6].tolist(), f["ch"][0]["e_a"].tolist()))
            if fr["ps"] is not None:
                p = fr["ps"][s]
                print("   ps: start %d is34 %d/%d ipdopd %d num_env %d/%d" % (p["start"], p["is34bands"], p["is34bands_old"],
                                                                         p["enable_ipdopd"], p["num_env"], p["num_env_old"]))
    # keep the two sides in step so that later frames are judged on their own
    d_state = torch.from_numpy(state).cuda()
print("streams that parted:", len(bad_streams))
