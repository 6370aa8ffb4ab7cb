"""Phase stamps of the fused HF + PS kernel (build with make EXTRA="-DHF_STAMPS -DPS_STAMPS").
N=<frames> python3 tools/hfps_stamps.py  -- one wave's last frame, in shader-clock ticks."""
import ctypes as C, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package(); synth = importlib.import_module("ffmpeg_heaac_amd.synth")
cfg = pkg.CFG_HEV2; n = int(os.environ.get("N", 2560))
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
hf = (C.c_ulonglong * 16)(); ps = (C.c_ulonglong * 16)()
pkg.lib().heaac_debug_hfps_stamps(hf); pkg.lib().heaac_debug_ps_stamps(ps)
h, p = list(hf), list(ps)
names = ["params", "lf_gen", "invfilt", "chirp", "consts", "mapping", "env_est", "gain", "assemble", "state"]
print("HF total", h[10] - h[0], " PS total", p[9] - p[0], " gap", p[0] - h[10])
for i in range(10):
    print("hf %-10s %8d" % (names[i], h[i + 1] - h[i]))
pn = ["cols+inb", "hybrid", "power", "transient", "H", "pass1", "pass2", "rest", "synth"]
for i in range(9):
    print("ps %-10s %8d" % (pn[i], p[i + 1] - p[i]))
