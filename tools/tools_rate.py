"""The spectral-tools kernel (noise substitution, AAC-Main prediction, M/S, intensity, TNS) by itself:
frames per second on the records of written access units, whole and with one tool at a time switched off in the
records, so that what each tool costs shows.  Records are parsed once on the host; the timed region is
heaac_spectral_tools_batch on device-resident records (HIP events).
usage: python tools/tools_rate.py [n frames] [stereo|mono] [lc|main]"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
pkg = importlib.import_module("ffmpeg-heaac_amd")
import test_parse as TP

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
cpe = (sys.argv[2] if len(sys.argv) > 2 else "stereo") == "stereo"
aot = 1 if (sys.argv[3] if len(sys.argv) > 3 else "lc") == "main" else 2
ch = 2 if cpe else 1
rng = np.random.default_rng(3)
base = 512
cfg = TP._cfg(pkg, aot, 3, ch)
aus = [TP._write_au(rng, 3, aot, cpe, extras=False, quiet=True)[0] for _ in range(base)]
st = np.zeros(base, pkg.AAC_STREAM_DT)
q = pkg.aac_parse_batch(cfg, st, aus, threads=4)
assert q["failed"] == 0
rep = (n + base - 1) // base
tools0 = np.tile(q["tools"], rep)[:n]
coeffs0 = np.tile(np.ascontiguousarray(q["coeffs"][:, :ch]), (rep, 1, 1))[:n]
dev = pkg.Device(n)


def variant(name):
    t = tools0.copy()
    if name == "no_tns":
        t["ch"]["tns"]["present"] = 0
    elif name == "no_pns":
        bt = t["ch"]["band_type"]
        bt[bt == 13] = 0
    elif name == "no_pred":
        t["ch"]["pred"]["predictor_present"] = 0
        t["ch"]["pred"]["pred_sfb_max"] = 0
        t["ch"]["pred"]["predictor_reset_group"] = 0
    elif name == "no_tns_no_pns":
        t["ch"]["tns"]["present"] = 0
        bt = t["ch"]["band_type"]
        bt[bt == 13] = 0
    elif name == "nothing":
        t[:] = np.zeros(1, t.dtype)
    return t


res = {}
for name in ("all", "no_tns", "no_pns", "no_pred", "no_tns_no_pns", "nothing"):
    d_t = pkg.to_device(variant(name))
    d_rng = torch.full((n,), 0x1f2e3d4c, dtype=torch.int32, device="cuda")
    d_pred = (torch.tensor([0, 0, 1, 1, 0, 0], dtype=torch.float32, device="cuda").repeat(n, ch * pkg.MAX_PREDICTORS, 1).contiguous()
              if aot == 1 else None)
    best = 1e9
    for it in range(4):
        d_c = torch.from_numpy(coeffs0).cuda()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dev.spectral_tools(ch, d_c, d_t, rng=d_rng, pred=d_pred)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    res[name] = best
frac_tns = float((tools0["ch"]["tns"]["present"][:, :ch] != 0).mean())
bytes_per_frame = ch * 8192 + (pkg.TOOLS_FRAME_DT.itemsize if cpe else 132 + 3500) + (ch * 672 * 24 * 2 if aot == 1 else 0)
out = dict(frames=n, channels=ch, object_type=aot, ms=res, frames_per_s=n / (res["all"] * 1e-3),
           algorithmic_bytes_per_frame=bytes_per_frame, hbm_frac=bytes_per_frame * n / (res["all"] * 1e-3) / 8e12,
           channels_with_tns=frac_tns)
print(json.dumps(out))
