"""Bitstream in, int16 PCM out, through host memory: what a caller that does NOT keep its data on the device gets
(PCIe-inclusive -- never the bench's `value`).  n HE-AACv2 streams, one access unit each per tick, through
include/heaac_pipeline.h: host parse (persistent pool) || H2D (pinned) || spectral tools + decode || D2H, consecutive
ticks overlapped.  Prints the stage times of a tick and the end-to-end rate with four ticks in flight, and the
rate when every tick is collected before the next is submitted (no overlap).
usage: python tools/e2e_rate.py [n streams] [ticks] [threads]"""
import ctypes as C, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
pkg = importlib.import_module("ffmpeg-heaac_amd")
import sbr_bitwriter as SW
import test_parse as TP
import test_sbr_parse as TS

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 12
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rng = np.random.default_rng(5)
cfg = TS._he_cfg(pkg, 1, True)
# 256 distinct streams written once, replicated over the batch (every stream still has its own parser state)
base = 256
writers = [SW.SbrStreamWriter(pkg, 1, ps=True, ps_modes="20") for _ in range(base)]
frames = []
for t in range(ticks):
    aus = []
    for w in writers:
        bits, _ = w.frame(rng)
        aus.append(TP._write_au(rng, 6, 2, False, extras=False, sbr=(bits, False), quiet=True)[0])
    keep = [C.create_string_buffer(a, len(a)) for a in aus]
    ptrs = (C.c_char_p * n)(*[C.cast(keep[i % base], C.c_char_p) for i in range(n)])
    sizes = (C.c_int * n)(*[len(aus[i % base]) for i in range(n)])
    frames.append((keep, ptrs, sizes, float(np.mean([len(a) for a in aus]))))


def run(overlap):
    pl = pkg.Pipeline(cfg, pkg.CFG_HEV2, n, threads=threads)
    stage = dict(parse=0.0, h2d=0.0, gpu=0.0, d2h=0.0)
    pl.submit_raw(frames[0][1], frames[0][2]); pl.collect()          # warm-up tick (first-touch, table upload)
    t0 = time.perf_counter()
    if overlap:
        depth, done = 4, 1
        for t in range(1, ticks):
            if t - done >= depth:
                pl.collect(); done += 1
                for k, v in pl.timing().items():
                    stage[k] += v
            pl.submit_raw(frames[t][1], frames[t][2])
        while done < ticks:
            pl.collect(); done += 1
            for k, v in pl.timing().items():
                stage[k] += v
    else:
        for t in range(1, ticks):
            pl.submit_raw(frames[t][1], frames[t][2])
            pl.collect()
            for k, v in pl.timing().items():
                stage[k] += v
    dt = time.perf_counter() - t0
    cnt = ticks - 1
    pl.close()
    return (ticks - 1) * n / dt, {k: v / cnt for k, v in stage.items()}


rate_seq, st_seq = run(False)
rate_ovl, st_ovl = run(True)
out = dict(streams=n, ticks=ticks, mean_access_unit_bytes=frames[0][3], host_threads=os.cpu_count(),
           stage_ms_per_tick=st_seq, stage_ms_per_tick_overlapped=st_ovl,
           frames_per_s_back_to_back=rate_seq, frames_per_s_overlapped=rate_ovl,
           note="PCIe-inclusive end-to-end rate of a host-buffer caller; not the bench metric")
print(json.dumps(out))
