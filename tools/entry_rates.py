"""Every batched entry point the bench does not time, by itself: time per call (HIP events), frames per second and the
fraction of 8 TB/s its own algorithmic bytes make -- a sweep for kernels nobody was looking at (the spectral tools were
at 2 - 3 % of the roofline until this way of looking found them: profiles/r04_experiments.md E8).
usage: python tools/entry_rates.py [n frames]"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
pkg = importlib.import_module("ffmpeg-heaac_amd")
synth = importlib.import_module("ffmpeg-heaac_amd.synth")
import bench as B

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = pkg.Device(n)


def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def line(name, ms, bytes_per_frame):
    print(json.dumps(dict(entry=name, frames=n, ms=round(ms, 4), frames_per_s=round(n / (ms * 1e-3)), bytes_per_frame=bytes_per_frame,
                          hbm_frac=round(bytes_per_frame * n / (ms * 1e-3) / 8e12, 3))), flush=True)


# ---- HE decode, every configuration and flag the bench leaves out
rng = np.random.default_rng(2)
hdr = synth.default_headers(pkg)
for cfg_name, down, fmt in (("CFG_HEV1_MONO", False, pkg.PCM_F32), ("CFG_HEV1", True, pkg.PCM_F32), ("CFG_HEV2", True, pkg.PCM_F32),
                            ("CFG_HEV1", False, pkg.PCM_S16), ("CFG_HEV1_MONO", False, pkg.PCM_S16)):
    cfg = getattr(pkg, cfg_name)
    pool = 2048
    frames = list(synth.he_stream(rng, cfg, pool, 2, hdr))
    reps = n // pool
    fr = frames[1]
    ch, words = pkg.CORE_CH[cfg], pkg.STATE_WORDS[cfg]
    d = dict(ics=pkg.to_device(fr["ics"]).repeat(reps).contiguous(), sbr=pkg.to_device(fr["sbr"]).repeat(reps).contiguous(),
             ps=pkg.to_device(fr["ps"]).repeat(reps).contiguous() if fr["ps"] is not None else None)
    coeffs = torch.from_numpy(np.ascontiguousarray(fr["coeffs"])).cuda().repeat(reps, 1, 1).contiguous()
    st = torch.zeros((n, words), device="cuda")
    st2 = torch.empty_like(st)
    d_hdr = pkg.to_device(hdr)
    length = 1024 if down else 2048
    pcm = (torch.empty((n, pkg.OUT_CH[cfg], length), device="cuda") if fmt == pkg.PCM_F32
           else torch.empty((n, length, pkg.OUT_CH[cfg]), dtype=torch.int16, device="cuda"))
    ms = timed(lambda: dev.he_decode(cfg, coeffs, d["ics"], d["sbr"], d_hdr, d["ps"], st, state_out=st2, pcm=pcm, pcm_format=fmt, downsampled=down))
    b = ch * 4096 + 2 * words * 4 + 680 + (532 if d["ps"] is not None else 0) + pkg.OUT_CH[cfg] * length * (4 if fmt == pkg.PCM_F32 else 2)
    line("he_decode %s%s %s" % (cfg_name[4:], " downsampled" if down else "", "f32" if fmt == pkg.PCM_F32 else "s16"), ms, b)

# ---- AAC-LC mono, and int16 output
for chn, fmt in ((1, pkg.PCM_F32), (2, pkg.PCM_S16), (1, pkg.PCM_S16)):
    coeffs_np, ics_np = next(iter(synth.lc_stream(rng, 2048, 1, channels=chn)))
    reps = n // 2048
    coeffs = torch.from_numpy(coeffs_np).cuda().repeat(reps, 1, 1).contiguous()
    ics = pkg.to_device(ics_np).repeat(reps).contiguous()
    st = torch.zeros((n, 512 * chn), device="cuda"); st2 = torch.empty_like(st)
    ms = timed(lambda: dev.lc_decode(chn, coeffs, ics, st, state_out=st2, pcm_format=fmt))
    line("lc_decode %d ch %s" % (chn, "f32" if fmt == pkg.PCM_F32 else "s16"), ms, chn * (4096 + 4096 + 1024 * (4 if fmt == pkg.PCM_F32 else 2)))

# ---- float planes -> interleaved int16 (5.1, 2048 samples), and independent coupling
planes = torch.rand((6, n, 2048), device="cuda") * 100 + 385
ms = timed(lambda: dev.pcm_interleave([(planes[c], 0, 2048) for c in range(6)], 2048))
line("pcm_interleave 6 x 2048", ms, 6 * 2048 * (4 + 2))
pcm = torch.rand((n, 2, 1024), device="cuda"); cce = torch.rand((n, 1024), device="cuda")
cpl = np.zeros(n, pkg.COUPLING_DT); cpl["on"] = 1; cpl["gain"] = 0.5
d_cpl = pkg.to_device(cpl)
ms = timed(lambda: dev.couple_after_imdct(2, pcm, cce, d_cpl))
line("couple_after_imdct 2 ch", ms, 2 * 4096 * 2 + 4096)
