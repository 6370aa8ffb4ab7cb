import sys, os, ctypes as C
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import importlib, numpy as np
pkg = importlib.import_module('ffmpeg-heaac_amd')
import test_parse as TP, test_pipeline as TPL, test_sbr_parse as TS
from test_damaged_streams_gpu import _mutate
from test_shim_gpu import HeaacCodecContext, HeaacPacket
mode = sys.argv[1]
lib = pkg.lib()
rng = np.random.default_rng(sum(map(ord, mode)))
n, ticks = 20, 9
cpe = mode == "hev1"; ch = 2 if cpe else 1
units = TPL._ticks(pkg, rng, ch, not cpe, n, ticks)
m4 = TS._he_cfg(pkg, ch, not cpe)
hcfg = pkg.CFG_HEV1 if cpe else pkg.CFG_HEV2
asc = bytes([0x2B, 0x11, 0x88, 0x00]) if cpe else bytes([0xEB, 0x09, 0x88, 0x00])
length, nout = 2048, 2
pool = [u for tick in units for u in tick]
fed = [list(t) for t in units]
for t in range(2, ticks):
    for i in range(n):
        if rng.random() < 0.4:
            fed[t][i] = _mutate(rng, fed[t][i], pool)
pl = pkg.Pipeline(m4, hcfg, n, threads=3)
got, status = [], []
for t in range(ticks):
    status.append(np.array(pl.submit(fed[t])).copy()); got.append(pl.collect().copy())
pl.close()
codec = C.c_void_p.in_dll(lib, "heaac_aac_decoder")
out = (C.c_int16 * (192000 // 2))()
for i in range(n):
    ctx = HeaacCodecContext(cfg=-1, extradata=asc, extradata_size=len(asc))
    assert lib.heaac_codec_open(C.byref(ctx), C.c_void_p(C.addressof(codec))) == 0
    hist = []
    for t in range(ticks):
        b = fed[t][i]
        buf = C.create_string_buffer(b, len(b)); pkt = HeaacPacket(C.cast(buf, C.c_void_p), len(b)); size = C.c_int(192000)
        used = lib.heaac_codec_decode(C.byref(ctx), out, C.byref(size), C.byref(pkt))
        same = None
        if used >= 0:
            pcm = np.frombuffer(out, np.int16, length * nout).reshape(length, nout)
            same = bool(np.array_equal(pcm, got[t][i]))
            if not same:
                d = np.argwhere(pcm != got[t][i])
                same = "first diff at %s of %d" % (d[0].tolist(), len(d))
        hist.append((t, used, int(status[t][i]), fed[t][i] != units[t][i], same))
    if any(h[4] not in (True, None) for h in hist):
        print("stream", i)
        for h in hist: print("   ", h)
        # the parser's view of the unit before the first mismatch
        st, sst, tab = np.zeros(1, pkg.AAC_STREAM_DT), pkg.sbr_streams(1), pkg.SbrHeaderTable(64)
        for t in range(ticks):
            p = pkg.heaac_parse_batch(m4, st, sst, tab, [fed[t][i]], with_ps=not cpe)
            print("    parse", t, int(p["status"][0]), dict(channels=int(p["info"]["channels"][0]), refused=int(p["info"]["refused"][0]), sbr_bit=int(p["info"]["sbr_payload_bit"][0]), start=int(p["sbr"]["start"][0])))
    lib.heaac_codec_close(C.byref(ctx))
