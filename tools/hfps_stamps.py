"""Phase timeline of the fused HF + PS kernel (build with make EXTRA="-DHEAAC_STAMPS": tools/build_variants.sh).
N=<frames> python3 tools/hfps_stamps.py  -- cycles per frame and phase, averaged over the frames of
wave 0 of every 8th workgroup, under full load."""
import ctypes as C, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package(); synth = importlib.import_module("ffmpeg_heaac_amd.synth")
cfg = pkg.CFG_HEV2; n = int(os.environ.get("N", 65536))
hdr = synth.default_headers(pkg); rng = np.random.default_rng(1)
dev = pkg.Device(n)
frames = list(synth.he_stream(rng, cfg, 256, 3, hdr))
rep = n // 256
st = torch.zeros((n, pkg.STATE_WORDS[cfg]), device="cuda")
d_hdr = pkg.to_device(hdr)
for fr in frames:
    args = (torch.from_numpy(fr["coeffs"]).cuda().repeat(rep, 1, 1), pkg.to_device(fr["ics"]).repeat(rep),
            pkg.to_device(fr["sbr"]).repeat(rep), d_hdr, pkg.to_device(fr["ps"]).repeat(rep))
    for _ in range(3):
        dev.he_decode(cfg, *args, st, state_out=st)
    torch.cuda.synchronize()
tl = (C.c_ulonglong * 33)()
assert pkg.lib().heaac_debug_timeline(tl) == 0
v = list(tl); cnt = max(1, v[32])
hn = ["params", "lf_gen", "invfilt", "chirp", "consts", "mapping", "env_est", "gain", "assemble", "state"]
pn = ["gap+p", "cols+inb", "hybrid", "power", "transient", "H", "pass1", "pass2", "rest", "synth"]
tot = sum(v[:32]) / cnt
print("frames %d  cycles/frame %.0f  between-frames %.0f" % (cnt, tot, v[31] / cnt))
print("HF: " + "  ".join("%s %.0f" % (hn[i], v[i + 1] / cnt) for i in range(10)))
print("PS: " + "  ".join("%s %.0f" % (pn[i], v[16 + i] / cnt) for i in range(10)))
sl = (C.c_ulonglong * 33)()
if hasattr(pkg.lib(), "heaac_debug_timeline_he") and pkg.lib().heaac_debug_timeline_he(sl) == 0:
    s = list(sl); c = max(1, s[32])
    sn = ["v_in", "imdct", "butterfly", "polyphase", "v_out"]
    print("SYN (per channel, %d): " % c + "  ".join("%s %.0f" % (sn[i], s[i + 1] / c) for i in range(5)))
