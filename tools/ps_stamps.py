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
st = torch.zeros((n, pkg.STATE_WORDS[cfg]), device="cuda"); st2 = st
d_hdr = pkg.to_device(hdr)
for fr in frames:
    args = (torch.from_numpy(fr["coeffs"]).cuda().repeat(rep, 1, 1), pkg.to_device(fr["ics"]).repeat(rep),
            pkg.to_device(fr["sbr"]).repeat(rep), d_hdr, pkg.to_device(fr["ps"]).repeat(rep))
    for _ in range(3):
        dev.he_decode(cfg, *args, st, state_out=st2)
    torch.cuda.synchronize()
out = (C.c_ulonglong * 16)()
pkg.lib().heaac_debug_ps_stamps(out)
v = list(out)
names = ["copy p", "tile+inb", "hybrid", "power", "transient", "H", "pass1", "pass2", "rest", "synth"]
print("total cycles", v[9] - v[0])
for i in range(9):
    print("%-10s %8d" % (names[i + 1] if i + 1 < len(names) else i, v[i + 1] - v[i]))
