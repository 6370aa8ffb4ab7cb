"""What damaged units cost the host-buffer pipeline: n HE-AACv2 streams, a share of them handed a unit that does not parse
in every tick (silence for the tick, state parked and put back).  ms per tick, ticks collected one by one.
usage: python tools/damage_rate.py [n streams] [ticks]"""
import ctypes as C, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
pkg = importlib.import_module("ffmpeg-heaac_amd")
import sbr_bitwriter as SW
import test_parse as TP
import test_sbr_parse as TS

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rng = np.random.default_rng(5)
cfg = TS._he_cfg(pkg, 1, True)
base = 256
writers = [SW.SbrStreamWriter(pkg, 1, ps=True, ps_modes="20") for _ in range(base)]
good = []
for t in range(ticks):
    aus = []
    for w in writers:
        bits, _ = w.frame(rng)
        aus.append(TP._write_au(rng, 6, 2, False, extras=False, sbr=(bits, False), quiet=True)[0])
    good.append(aus)
bad = C.create_string_buffer(bytes([0x40, 0]) + bytes(8), 10)      # a coupling element: refused, nothing moves
res = {}
for share in (0.0, 0.001, 0.01, 0.1):
    failing = rng.random(n) < share
    pl = pkg.Pipeline(cfg, pkg.CFG_HEV2, n, threads=0)
    frames = []
    for t in range(ticks):
        keep = [C.create_string_buffer(a, len(a)) for a in good[t]]
        ptrs = (C.c_char_p * n)(*[C.cast(bad if (failing[i] and t > 0) else keep[i % base], C.c_char_p) for i in range(n)])
        sizes = (C.c_int * n)(*[10 if (failing[i] and t > 0) else len(good[t][i % base]) for i in range(n)])
        frames.append((keep, ptrs, sizes))
    pl.submit_raw(frames[0][1], frames[0][2]); pl.collect()
    t0 = time.perf_counter()
    for t in range(1, ticks):
        pl.submit_raw(frames[t][1], frames[t][2]); pl.collect()
    res[share] = (time.perf_counter() - t0) / (ticks - 1) * 1e3
    pl.close()
print(json.dumps(dict(streams=n, ms_per_tick_by_failing_share={str(k): round(v, 2) for k, v in res.items()})))
