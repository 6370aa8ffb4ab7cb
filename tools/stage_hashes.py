"""Per-frame stage hashes of one stored bitstream through parser + oracle (no GPU): to compare two machines.
usage: python tools/stage_hashes.py <name>"""
import hashlib, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
pkg = importlib.import_module("ffmpeg-heaac_amd")
import oracle_lib as oracle
import make_bitstream_vectors as B
name = sys.argv[1]
v = json.load(open(os.path.join(ROOT, "tests", "golden", "bitstreams.json")))[name]
asc, si, cpe, sbr, ps, frames, seed = B.STREAMS[name]
ch = 2 if cpe else 1
m4, _ = pkg.asc_parse(asc)
if sbr: m4.sbr = 1
if ps: m4.ps = 1
hcfg = (pkg.CFG_HEV1 if cpe else pkg.CFG_HEV2) if sbr else (pkg.CFG_LC_STEREO if cpe else pkg.CFG_LC_MONO)
tab = pkg.SbrHeaderTable(64)
st, sst = np.zeros(1, pkg.AAC_STREAM_DT), pkg.sbr_streams(1)
state = np.zeros((1, pkg.STATE_WORDS[hcfg]), np.float32)
rs = np.full(1, 0x1f2e3d4c, np.int32)
h = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:10]
print("tables", h(pkg.get_table("kbd_long")), h(pkg.get_table("sine_long")), h(pkg.get_table("qmf_ds")))
for t, a in enumerate(v["access_units"]):
    au = bytes.fromhex(a)
    p = pkg.heaac_parse_batch(m4, st, sst, tab, [au], threads=1, with_ps=ps) if sbr else pkg.aac_parse_batch(m4, st, [au], threads=1)
    coeffs = np.ascontiguousarray(p["coeffs"][:, :ch])
    c, rs = oracle.spectral_tools_batch(ch, coeffs, p["tools"], rng=rs)
    ics = np.ascontiguousarray(p["ics"][:, :ch])
    if sbr:
        f32, _ = oracle.he_decode_batch(hcfg, c, ics, p["sbr"], tab.headers(), p["ps"] if ps else None, state, oracle.PCM_F32)
        ref, state = oracle.he_decode_batch(hcfg, c, ics, p["sbr"], tab.headers(), p["ps"] if ps else None, state, oracle.PCM_S16)
    else:
        f32, _ = oracle.lc_decode_batch(ch, c, ics, state, oracle.PCM_F32)
        ref, state = oracle.lc_decode_batch(ch, c, ics, state, oracle.PCM_S16)
    print(t, "coeffs", h(coeffs), "tools", h(p["tools"]), "after", h(c), "f32", h(f32), "s16", h(ref), "state", h(state),
          "hdr", h(tab.headers()), "sbr", h(p["sbr"]) if sbr else "-")
