"""Bitstream in, int16 PCM out, through host memory: what a caller that does NOT keep its data on the device gets.
n HE-AACv2 streams, one access unit each per tick: host parse (all cores) -> H2D (pinned) -> spectral tools +
decode on the GPU -> D2H.  Prints the rate of every stage and of the whole tick run back to back (no overlap).
usage: python tools/e2e_rate.py [n streams] [ticks]"""
import ctypes as C, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
pkg = importlib.import_module("ffmpeg-heaac_amd")
import sbr_bitwriter as SW
import test_parse as TP
import test_sbr_parse as TS

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 6
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
    frames.append([aus[i % base] for i in range(n)])
L = pkg.lib()
tab = pkg.SbrHeaderTable(4096)
st = np.zeros(n, pkg.AAC_STREAM_DT); sst = pkg.sbr_streams(n)
pin = lambda a: torch.from_numpy(a).pin_memory()
h_coeffs = pin(np.zeros((n, 2, 1024), np.float32)); h_ics = pin(np.zeros((n, 2, 4), np.uint8))
h_tools = pin(np.zeros((n, pkg.TOOLS_FRAME_DT.itemsize), np.uint8)); h_sbr = pin(np.zeros((n, pkg.SBR_FRAME_DT.itemsize), np.uint8))
h_ps = pin(np.zeros((n, pkg.PS_FRAME_DT.itemsize), np.uint8)); h_pcm = pin(np.zeros((n, 2048, 2), np.int16))
status = np.zeros(n, np.int32)
dev = pkg.Device(n)
d_state = torch.zeros((n, pkg.STATE_WORDS[pkg.CFG_HEV2]), device="cuda")
d_rng = torch.full((n,), 0x1f2e3d4c, dtype=torch.int32, device="cuda")
p_ = lambda t: C.c_void_p(t.data_ptr())
tm = dict(parse=0.0, h2d=0.0, gpu=0.0, d2h=0.0)
for t, aus in enumerate(frames):
    keep = [C.create_string_buffer(a, len(a)) for a in aus]
    ptrs = (C.c_char_p * n)(*[C.cast(k, C.c_char_p) for k in keep]); sizes = (C.c_int * n)(*[len(a) for a in aus])
    t0 = time.perf_counter()
    failed = L.heaac_heaac_parse_batch(C.byref(cfg), st.ctypes.data_as(C.c_void_p), sst.ctypes.data_as(C.c_void_p), C.c_void_p(tab._h),
                                       ptrs, sizes, C.c_size_t(n), p_(h_coeffs), p_(h_ics), p_(h_tools), p_(h_sbr), p_(h_ps),
                                       None, status.ctypes.data_as(C.c_void_p), C.c_int(0))
    t1 = time.perf_counter()
    assert failed == 0
    # the mono spectrum is channel 0 of each [2][1024] pair: copy the strided view (the device wants [n][1][1024])
    d_c = h_coeffs[:, 0].contiguous().cuda(non_blocking=True) if False else h_coeffs.cuda(non_blocking=True)[:, 0].contiguous()
    d_ics = h_ics.cuda(non_blocking=True)[:, 0].contiguous(); d_tools = h_tools.cuda(non_blocking=True)
    d_sbr = h_sbr.cuda(non_blocking=True); d_ps = h_ps.cuda(non_blocking=True)
    d_hdr = pkg.to_device(tab.headers())
    torch.cuda.synchronize(); t2 = time.perf_counter()
    dev.spectral_tools(1, d_c, d_tools, rng=d_rng)
    pcm, d_state = dev.he_decode(pkg.CFG_HEV2, d_c, d_ics, d_sbr, d_hdr, d_ps, d_state, state_out=d_state, pcm_format=pkg.PCM_S16)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    h_pcm.copy_(pcm, non_blocking=True); torch.cuda.synchronize(); t4 = time.perf_counter()
    if t:                                             # the first tick warms everything up
        tm["parse"] += t1 - t0; tm["h2d"] += t2 - t1; tm["gpu"] += t3 - t2; tm["d2h"] += t4 - t3
k = (ticks - 1) * n
print("streams %d, ticks %d (first one not timed), mean access unit %.0f bytes, %d host threads" %
      (n, ticks, np.mean([len(a) for a in frames[0]]), os.cpu_count()))
for name in ("parse", "h2d", "gpu", "d2h"):
    print("  %-6s %8.2f ms per tick   %9.3f M frames/s" % (name, 1e3 * tm[name] / (ticks - 1), k / tm[name] / 1e6))
tot = sum(tm.values())
print("  whole tick, stages back to back: %.3f M frames/s" % (k / tot / 1e6))
