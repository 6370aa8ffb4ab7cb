"""Decode one stored bitstream (tests/golden/bitstreams.json) through the codec surface and through parser + oracle,
frame by frame; report the first difference.  usage: python tools/debug_golden_stream.py <name>"""
import ctypes as C, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
pkg = importlib.import_module("ffmpeg-heaac_amd")
import oracle_lib as oracle
import make_bitstream_vectors as B
from test_shim_gpu import HeaacCodecContext, HeaacPacket
name = sys.argv[1]
v = json.load(open(os.path.join(ROOT, "tests", "golden", "bitstreams.json")))[name]
asc, si, cpe, sbr, ps, frames, seed = B.STREAMS[name]
ch = 2 if cpe else 1
lib = pkg.lib()
ctx = HeaacCodecContext(cfg=-1, extradata=asc, extradata_size=len(asc))
codec = C.c_void_p.in_dll(lib, "heaac_aac_decoder")
assert lib.heaac_codec_open(C.byref(ctx), C.c_void_p(C.addressof(codec))) == 0
m4, _ = pkg.asc_parse(asc)
if sbr: m4.sbr = 1
if ps: m4.ps = 1
hcfg = (pkg.CFG_HEV1 if cpe else pkg.CFG_HEV2) if sbr else (pkg.CFG_LC_STEREO if cpe else pkg.CFG_LC_MONO)
tab = pkg.SbrHeaderTable(64)
st, sst = np.zeros(1, pkg.AAC_STREAM_DT), pkg.sbr_streams(1)
state = np.zeros((1, pkg.STATE_WORDS[hcfg]), np.float32)
rs = np.full(1, 0x1f2e3d4c, np.int32)
out = (C.c_int16 * (192000 // 2))()
for t, a in enumerate(v["access_units"]):
    au = bytes.fromhex(a)
    buf = C.create_string_buffer(au, len(au)); pkt = HeaacPacket(C.cast(buf, C.c_void_p), len(au)); size = C.c_int(192000)
    used = lib.heaac_codec_decode(C.byref(ctx), out, C.byref(size), C.byref(pkt))
    got = np.frombuffer(out, np.int16, size.value // 2).reshape(-1, pkg.OUT_CH[hcfg]).copy()
    p = pkg.heaac_parse_batch(m4, st, sst, tab, [au], threads=1, with_ps=ps) if sbr else pkg.aac_parse_batch(m4, st, [au], threads=1)
    coeffs = np.ascontiguousarray(p["coeffs"][:, :ch])
    c, rs = oracle.spectral_tools_batch(ch, coeffs, p["tools"], rng=rs)
    ics = np.ascontiguousarray(p["ics"][:, :ch])
    if sbr:
        ref, state = oracle.he_decode_batch(hcfg, c, ics, p["sbr"], tab.headers(), p["ps"] if ps else None, state, oracle.PCM_S16)
        reff, _ = oracle.he_decode_batch(hcfg, c, ics, p["sbr"], tab.headers(), p["ps"] if ps else None, state * 0, oracle.PCM_F32)
    else:
        ref, state = oracle.lc_decode_batch(ch, c, ics, state, oracle.PCM_S16)
    bad = got != ref[0]
    f = p["sbr"][0] if sbr else None
    print("frame", t, "used", used, "status", p["status"], "bad", int(bad.sum()), "max|pcm|", int(np.abs(ref).max()),
          "finite" if not sbr else bool(np.isfinite(reff).all()),
          "" if f is None else dict(start=int(f["start"]), reset=int(f["reset"]), hdr=int(f["hdr"]), cpl=int(f["bs_coupling"]),
                                    L=[int(f["ch"][c_]["bs_num_env"]) for c_ in range(ch)]))
    if bad.any():
        w = np.argwhere(bad)
        print("   first", w[:4].tolist(), "got", got[bad][:4].tolist(), "ref", ref[0][bad][:4].tolist())
