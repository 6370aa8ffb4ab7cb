"""Per-call (batch of one) rate of the AVCodec-shaped surface: packets of one stream through
heaac_codec_decode (H2D, kernels, D2H per call).  Prints frames/s per context -- the latency-bound
number INTEGRATION.md s2 refers to; the throughput path is the batched ABI."""
import ctypes as C, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
from test_shim_gpu import HeaacCodecContext, HeaacPacket
pkg = g.load_package(); synth = importlib.import_module(g.PKG_NAME + ".synth")
lib = pkg.lib()
for cfgname in ("CFG_LC_STEREO", "CFG_HEV2"):
    cfg = getattr(pkg, cfgname)
    rng = np.random.default_rng(1)
    ctx = HeaacCodecContext(cfg=cfg)
    codec = C.c_void_p.in_dll(lib, "heaac_aac_decoder")
    assert lib.heaac_codec_open(C.byref(ctx), C.c_void_p(C.addressof(codec))) == 0
    hdr = synth.default_headers(pkg)
    steps = 300
    if cfg == pkg.CFG_LC_STEREO:
        frames = [dict(coeffs=c, ics=i, sbr=None, ps=None) for c, i in synth.lc_stream(rng, 1, steps, 2)]
    else:
        frames = list(synth.he_stream(rng, cfg, 1, steps, hdr))
    pkts = []
    for t, fr in enumerate(frames):
        ics2 = np.zeros(2, pkg.ICS_DT); ics2[: pkg.CORE_CH[cfg]] = fr["ics"][0]
        head = np.zeros(1, np.dtype([("magic", "<u4"), ("cfg", "<u2"), ("flags", "<u2"), ("ics", pkg.ICS_DT, (2,))]))
        head["magic"] = 0x48454141; head["cfg"] = cfg; head["ics"][0] = ics2
        blob = head.tobytes() + fr["coeffs"][0].astype(np.float32).tobytes()
        if fr["sbr"] is not None:
            head["flags"] = 1 if t == 0 else 0
            blob = head.tobytes() + fr["coeffs"][0].astype(np.float32).tobytes() + fr["sbr"][0].tobytes()
            if fr["ps"] is not None: blob += fr["ps"][0].tobytes()
            if t == 0: blob += hdr[0].tobytes()
        pkts.append(C.create_string_buffer(blob, len(blob)))
    out = (C.c_int16 * (192000 // 2))()
    def run(i):
        pkt = HeaacPacket(C.cast(pkts[i], C.c_void_p), len(pkts[i])); size = C.c_int(192000)
        assert lib.heaac_codec_decode(C.byref(ctx), out, C.byref(size), C.byref(pkt)) == len(pkts[i])
    for i in range(20): run(i)
    t0 = time.perf_counter()
    for i in range(20, steps): run(i)
    dt = time.perf_counter() - t0
    print("%s: %.0f frames/s per context (%.1f us per call)" % (cfgname, (steps - 20) / dt, dt / (steps - 20) * 1e6))
    lib.heaac_codec_close(C.byref(ctx))
