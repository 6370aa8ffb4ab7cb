"""Mint tests/golden/pcm_sha256.json: SHA-256 of the oracle's PCM and state for seeded synthetic
streams (the inputs are regenerated from the seed by tests/test_golden.py, so only hashes are
stored).  The reference cannot be built here, so these vectors pin the ORACLE against
drift; the GPU test checks the product against the same hashes without calling the oracle.

    python tests/golden/make_pcm_checksums.py
"""
import hashlib, importlib, json, os, sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = {
    # name: (kind, cfg attr, frames, steps, seed, ps_mode)
    "lc_stereo": ("lc", None, 24, 4, 501, None),
    "lc_mono": ("lc1", None, 17, 4, 502, None),
    "hev1": ("he", "CFG_HEV1", 12, 4, 503, None),
    "hev1_mono": ("he", "CFG_HEV1_MONO", 9, 3, 504, None),
    "hev2_20": ("he", "CFG_HEV2", 12, 4, 505, "20"),
    "hev2_mix": ("he", "CFG_HEV2", 10, 5, 506, "mix"),
    "tools_cpe": ("tools", None, 40, 2, 507, None),
    # the decoder's degrade and transition paths (synth.he_stream events): frames before the first SBR
    # header, mid-stream header changes, unusable SBR / PS payloads
    "hev1_events": ("he_ev", "CFG_HEV1", 14, 7, 508, None),
    "hev2_events": ("he_ev", "CFG_HEV2", 14, 7, 509, "mix"),
}
EVENTS = dict(lead_in=2, p_switch=0.3, p_drop=0.15, p_ps_off=0.2)


def _h(*arrays):
    m = hashlib.sha256()
    for a in arrays:
        m.update(np.ascontiguousarray(a).tobytes())
    return m.hexdigest()


def run_case(name, pkg, synth, decode_lc, decode_he, tools):
    """decode_*: callables with the oracle's signatures; returns list of per-step hashes."""
    kind, cfg_name, n, steps, seed, ps_mode = CASES[name]
    rng = np.random.default_rng(seed)
    out = []
    if kind in ("lc", "lc1"):
        ch = 2 if kind == "lc" else 1
        state = np.zeros((n, ch * 512), np.float32)
        for coeffs, ics in synth.lc_stream(rng, n, steps, ch):
            pcm, state = decode_lc(ch, coeffs, ics, state, pkg.PCM_S16)
            out.append(_h(pcm, state))
    elif kind in ("he", "he_ev"):
        cfg = getattr(pkg, cfg_name)
        hdr = synth.default_headers(pkg, extra=True, null=kind == "he_ev")
        state = np.zeros((n, pkg.STATE_WORDS[cfg]), np.float32)
        kw = dict(ps_mode=ps_mode) if ps_mode else {}
        if kind == "he_ev":
            kw.update(events=EVENTS, coupling=0.3)
        nh = len(hdr) - (kind == "he_ev")
        for fr in synth.he_stream(rng, cfg, n, steps, hdr, hdr_choice=np.arange(n) % nh, **kw):
            pcm, state = decode_he(cfg, fr["coeffs"], fr["ics"], fr["sbr"], hdr, fr["ps"], state, pkg.PCM_S16)
            out.append(_h(pcm, state))
    else:
        rs = np.full(n, 0x1f2e3d4c, np.int32)
        for _ in range(steps):
            t = synth.tools_frames(rng, pkg, n, 2)
            c = (rng.standard_normal((n, 2, 1024)) * 1e-4).astype(np.float32)
            c, rs = tools(2, c, t, rs)
            out.append(_h(c, rs))
    return out


if __name__ == "__main__":
    import __graft_entry__ as g
    import oracle_lib as oracle
    pkg = g.load_package()
    synth = importlib.import_module(g.PKG_NAME + ".synth")
    res = {k: run_case(k, pkg, synth, oracle.lc_decode_batch, oracle.he_decode_batch, oracle.spectral_tools_batch)
           for k in CASES}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pcm_sha256.json")
    json.dump(res, open(path, "w"), indent=1)
    print("wrote", path)
