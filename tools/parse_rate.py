"""Host parser throughput: whole HE-AAC access units (core element + SBR payload with PS) per second through
heaac_heaac_parse_batch, on 1 thread and on all cores.  Usage: python tools/parse_rate.py [n] """
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
pkg = importlib.import_module("ffmpeg-heaac_amd")
import sbr_bitwriter as SW
import test_parse as TP
import test_sbr_parse as TS

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rng = np.random.default_rng(1)
cfg = TS._he_cfg(pkg, 1, True)
writers = [SW.SbrStreamWriter(pkg, 1, ps=True, ps_modes="20") for _ in range(n)]
frames = []
for step in range(3):
    aus = []
    for w in writers:
        bits, _ = w.frame(rng)
        aus.append(TP._write_au(rng, 6, 2, False, extras=False, sbr=(bits, False), quiet=True)[0])
    frames.append(aus)
print("mean access unit: %.0f bytes" % np.mean([len(a) for f in frames for a in f]))
import ctypes as C
L = pkg.lib()
marsh = []
for aus in frames:
    keep = [C.create_string_buffer(bytes(a), len(a)) for a in aus]
    marsh.append((keep, (C.c_char_p * n)(*[C.cast(k, C.c_char_p) for k in keep]), (C.c_int * n)(*[len(a) for a in aus])))
coeffs = np.zeros((n, 2, 1024), np.float32); ics = np.zeros((n, 2), pkg.ICS_DT); tools = np.zeros(n, pkg.TOOLS_FRAME_DT)
sbr = np.zeros(n, pkg.SBR_FRAME_DT); ps = np.zeros(n, pkg.PS_FRAME_DT); status = np.zeros(n, np.int32)
p_ = lambda a: a.ctypes.data_as(C.c_void_p)
for threads in (1, 2, 4, os.cpu_count()):
    best = 0.0
    for rep in range(5):
        tab = pkg.SbrHeaderTable(4096)
        st = np.zeros(n, pkg.AAC_STREAM_DT); sst = pkg.sbr_streams(n)
        t0 = time.perf_counter()
        for keep, ptrs, sizes in marsh:
            failed = L.heaac_heaac_parse_batch(C.byref(cfg), p_(st), p_(sst), C.c_void_p(tab._h), ptrs, sizes, C.c_size_t(n),
                                               p_(coeffs), p_(ics), p_(tools), p_(sbr), p_(ps), None, p_(status), C.c_int(threads))
            assert failed == 0
        dt = time.perf_counter() - t0
        best = max(best, len(marsh) * n / dt)
    print("threads %2d: %.0f access units/s" % (threads, best))
