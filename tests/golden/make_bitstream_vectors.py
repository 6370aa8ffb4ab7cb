"""Mint tests/golden/bitstreams.json: short AAC / HE-AAC streams as BYTES (access units written by the test
bit writers, hex), their AudioSpecificConfig, and what must come out of them -- SHA-256 of the parsed SBR / PS
records per frame and of the int16 PCM of the whole stream (host parser -> oracle spectral tools -> oracle decode).
The vectors are data: once minted they pin parser + decoder against drift without the writers' models, and
`tests/test_golden.py` decodes the same bytes through the codec surface on the GPU.

    python tests/golden/make_bitstream_vectors.py            # write new streams
    python tests/golden/make_bitstream_vectors.py --rehash   # keep the stored BYTES, re-derive what must come out of them
                                                             # (after a deliberate change of parser semantics)
"""
import hashlib, importlib, json, os, sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

STREAMS = {
    # name: (asc bytes, sampling index, cpe, sbr writer?, ps?, frames, seed)
    "lc_stereo_48k": (bytes([0x11, 0x90]), 3, True, False, False, 8, 801),
    "hev1_stereo_24k": (bytes([0x2B, 0x11, 0x88, 0x00]), 6, True, True, False, 8, 802),
    "hev2_mono_24k": (bytes([0xEB, 0x09, 0x88, 0x00]), 6, False, True, True, 10, 803),
    "hev2_implicit_24k": (bytes([0x13, 0x08]), 6, False, True, True, 6, 804),
}


# multi-element layouts (tests/test_parse_layout.py writes the units; elements in bitstream order)
SCE, CPE, LFE = 0, 1, 3
LAYOUT_STREAMS = {
    # name: (asc bytes, sampling index, channel configuration, elements, SBR, frames, seed)
    "lc_5_1_48k": (bytes([0x11, 0xB0]), 3, 6, [(SCE, 0), (CPE, 0), (CPE, 1), (LFE, 0)], False, 6, 811),
    "hev1_5_1_24k": (bytes([0x2B, 0x31, 0x88, 0x00]), 6, 6, [(SCE, 0), (CPE, 0), (CPE, 1), (LFE, 0)], True, 6, 812),
}


# streams whose program config element names coupling channel elements (tests/coupled_ref.py writes and checks them)
COUPLED_STREAMS = {
    # name: (object type, sampling index, output elements, coupling element tags, coupling points, frames, seed)
    "lc_pce_5_1_coupled_48k": (2, 3, [(SCE, 0), (CPE, 0), (CPE, 1), (LFE, 0)], [3, 9], [0, 1, 3], 6, 821),
    "main_pce_pair_coupled_48k": (1, 3, [(CPE, 2)], [6], [0, 1, 3], 6, 822),
    "hev1_pce_three_coupled_24k": (2, 6, [(SCE, 0), (CPE, 0), (LFE, 1)], [4, 11], [0, 1, 3], 6, 823),   # SBR: si = 6
    # four coupling elements on one pair (round 4: HEAAC_MAX_CCE covers every instance tag; the reference has no limit)
    "lc_pce_pair_four_coupled_48k": (2, 3, [(CPE, 1)], [0, 5, 9, 15], [0, 1, 3], 5, 824),
}


def write_coupled_stream(pkg, oracle, name):
    """Returns (AudioSpecificConfig bytes, access units)."""
    import coupled_ref as R
    aot, si, elems, cc_tags, points, frames, seed = COUPLED_STREAMS[name]
    he = si == 6
    rng = np.random.default_rng(seed)
    asc = R.asc(aot, si, elems, cc_tags, rng, he=he)
    r, m4, layout = pkg.asc_layout(asc)
    assert r == 0
    chk = R.Checker(pkg, oracle, m4, layout, aot, he=he)
    uw = R.UnitWriter(pkg, rng, si, aot, elems, cc_tags, points, he, quiet=True)
    aus = []
    for t in range(frames):
        aus.append(uw.unit(cc_tags, chk.parses, pts=[3] if 3 in points and t in (1, 4) else None))
        chk.frame(aus[-1])
    return asc, aus


def decode_coupled_stream(pkg, oracle, name, asc, aus):
    import coupled_ref as R
    aot, si = COUPLED_STREAMS[name][:2]
    r, m4, layout = pkg.asc_layout(asc)
    assert r == 0
    chk = R.Checker(pkg, oracle, m4, layout, aot, he=si == 6)
    rec, pcm = [], hashlib.sha256()
    for au in aus:
        out, g = chk.frame(au)
        m = hashlib.sha256(g["tools"].tobytes())
        for k in ("elem", "cce", "cce_tools", "cce_ics", "cce_elem"):
            m.update(g[k].tobytes())
        rec.append(m.hexdigest())
        pcm.update(out.tobytes())
    # both kinds of coupling took place, and a vector must not depend on how a machine turns NaN into int16
    assert chk.dependent and chk.independent and chk.finite, name
    return rec, pcm.hexdigest(), list(out.shape)


def write_layout_stream(pkg, name):
    import copy
    import sbr_bitwriter as SW
    import test_parse_layout as TL
    asc, si, cc, elems, he, frames, seed = LAYOUT_STREAMS[name]
    rng = np.random.default_rng(seed)
    writers = {k: SW.SbrStreamWriter(pkg, 2 if t == CPE else 1) for k, (t, _) in enumerate(elems) if t != LFE}
    aus = []
    for t in range(frames):
        payloads = None
        if he:
            payloads = []
            for k, (typ, _) in enumerate(elems):
                if typ == LFE:
                    payloads.append(None)
                    continue
                w = writers[k]
                while True:
                    keep = copy.deepcopy((w.ch, w.ps, w.header, w.hdr_rec, w.kx_m, w.coupling))
                    bits, _ = w.frame(rng, new_header=(t == frames // 2 and k == 1), respec=(t == frames // 2 and k == 1))
                    if (4 + len(bits) + 7) // 8 <= 269:
                        break
                    w.ch, w.ps, w.header, w.hdr_rec, w.kx_m, w.coupling = keep
                payloads.append(bits)
        aus.append(TL.build(rng, si, 2, elems, extras=bool(t & 1), payloads=payloads)[0])
    return aus


def decode_layout_stream(pkg, oracle, name, aus):
    """Layout parser + oracle per element + ff_float_to_int16_interleave_c; returns (record hashes, PCM hash, shape)."""
    asc, si, cc, elems, he, frames, seed = LAYOUT_STREAMS[name]
    r, m4, layout = pkg.asc_layout(asc)
    assert r == 0 and m4.chan_config == cc
    ne, nch = int(layout[0]["n_elements"]), int(layout[0]["channels"])
    slot_ch = [int(layout[0]["elem"][e]["channels"]) for e in range(ne)]
    st = np.zeros(pkg.MAX_ELEMENTS, pkg.AAC_STREAM_DT)
    state = [np.zeros((1, pkg.STATE_WORDS[(pkg.CFG_HEV1 if c == 2 else pkg.CFG_HEV1_MONO) if he else
                                          (pkg.CFG_LC_STEREO if c == 2 else pkg.CFG_LC_MONO)]), np.float32) for c in slot_ch]
    rng_state = np.full(1, 0x1f2e3d4c, np.int32)
    tab, sst = pkg.SbrHeaderTable(64), pkg.sbr_streams(ne)
    rec, pcm = [], hashlib.sha256()
    for au in aus:
        r, p = pkg.aac_parse_frame_layout(m4, layout, st, au)
        assert r == 0
        m = hashlib.sha256(p["tools"].tobytes())
        m.update(p["elem"].tobytes())
        spec = [None] * ne
        for e in sorted(range(ne), key=lambda e: int(p["elem"][e]["seq"])):
            c = slot_ch[e]
            spec[e], rng_state = oracle.spectral_tools_batch(c, np.ascontiguousarray(p["coeffs"][e:e + 1, :c]),
                                                             p["tools"][e:e + 1], rng=rng_state)
        planes = [None] * nch
        for e in range(ne):
            c = slot_ch[e]
            ics = np.ascontiguousarray(p["ics"][e:e + 1, :c])
            if he:
                ei = p["elem"][e]
                if int(ei["sbr_payload_bit"]) >= 0:
                    rr, sbr, _, _ = pkg.sbr_parse_payload(sst[e], tab, m4.sample_rate, au, c, False,
                                                          bit=int(ei["sbr_payload_bit"]), cnt=int(ei["sbr_payload_bytes"]))
                    assert rr == 0
                else:
                    sbr = pkg.sbr_no_payload(sst[e], c)
                m.update(sbr.tobytes())
                m.update(tab.headers()[int(sbr["hdr"][0])].tobytes())
                f32, state[e] = oracle.he_decode_batch(pkg.CFG_HEV1 if c == 2 else pkg.CFG_HEV1_MONO, spec[e], ics, sbr,
                                                       tab.headers(), None, state[e], oracle.PCM_F32)
            else:
                f32, state[e] = oracle.lc_decode_batch(c, spec[e], ics, state[e], oracle.PCM_F32)
            assert np.isfinite(f32).all() and np.abs(f32 - 385.0).max() < 4.0, name
            for j in range(c):
                planes[int(layout[0]["elem"][e]["first_channel"]) + j] = f32[0, j]
        out = oracle.float_to_int16_interleave(planes)
        rec.append(m.hexdigest())
        pcm.update(out.tobytes())
    return rec, pcm.hexdigest(), list(out.shape)


def write_stream(pkg, name):
    import copy
    import sbr_bitwriter as SW
    import test_parse as TP
    asc, si, cpe, sbr, ps, frames, seed = STREAMS[name]
    rng = np.random.default_rng(seed)
    w = SW.SbrStreamWriter(pkg, 2 if cpe else 1, ps=ps) if sbr else None
    aus = []
    for t in range(frames):
        payload = None
        if w is not None:
            while True:
                keep = copy.deepcopy((w.ch, w.ps, w.header, w.hdr_rec, w.kx_m, w.coupling))
                bits, _ = w.frame(rng, new_header=t == frames // 2, respec=t == frames // 2)
                if (4 + len(bits) + 7) // 8 <= 269:
                    break
                w.ch, w.ps, w.header, w.hdr_rec, w.kx_m, w.coupling = keep
            payload = (bits, False)
        aus.append(TP._write_au(rng, si, 2, cpe, extras=False, sbr=payload, quiet=True)[0])
    return aus


def decode_stream(pkg, oracle, name, aus):
    """Parser + oracle; returns (per-frame record hashes, PCM hash, out shape)."""
    asc, si, cpe, sbr, ps, frames, seed = STREAMS[name]
    ch = 2 if cpe else 1
    m4, _ = pkg.asc_parse(asc)
    if sbr:
        m4.sbr = 1
    if ps:
        m4.ps = 1
    hcfg = (pkg.CFG_HEV1 if cpe else pkg.CFG_HEV2) if sbr else (pkg.CFG_LC_STEREO if cpe else pkg.CFG_LC_MONO)
    tab = pkg.SbrHeaderTable(64)
    st, sst = np.zeros(1, pkg.AAC_STREAM_DT), pkg.sbr_streams(1)
    state = np.zeros((1, pkg.STATE_WORDS[hcfg]), np.float32)
    rng_state = np.full(1, 0x1f2e3d4c, np.int32)
    rec, pcm = [], hashlib.sha256()
    for au in aus:
        if sbr:
            p = pkg.heaac_parse_batch(m4, st, sst, tab, [au], threads=1, with_ps=ps)
            m = hashlib.sha256(p["sbr"].tobytes())
            m.update(tab.headers()[int(p["sbr"]["hdr"][0])].tobytes())
            if ps:
                m.update(p["ps"].tobytes())
            rec.append(m.hexdigest())
        else:
            p = pkg.aac_parse_batch(m4, st, [au], threads=1)
            rec.append(hashlib.sha256(p["tools"].tobytes()).hexdigest())
        assert p["failed"] == 0
        coeffs = np.ascontiguousarray(p["coeffs"][:, :ch])
        c, rng_state = oracle.spectral_tools_batch(ch, coeffs, p["tools"], rng=rng_state)
        ics = np.ascontiguousarray(p["ics"][:, :ch])
        if sbr:
            f32, _ = oracle.he_decode_batch(hcfg, c, ics, p["sbr"], tab.headers(), p["ps"] if ps else None, state,
                                            oracle.PCM_F32)
            # a vector must not depend on how a machine encodes NaN: finite and inside the int16 range
            assert np.isfinite(f32).all() and np.abs(f32 - 385.0).max() < 4.0, name
            out, state = oracle.he_decode_batch(hcfg, c, ics, p["sbr"], tab.headers(), p["ps"] if ps else None, state,
                                                oracle.PCM_S16)
        else:
            out, state = oracle.lc_decode_batch(ch, c, ics, state, oracle.PCM_S16)
        pcm.update(out.tobytes())
    return rec, pcm.hexdigest(), list(out.shape[1:])


def main():
    pkg = importlib.import_module("ffmpeg-heaac_amd")
    import oracle_lib as oracle
    out = {}
    path = os.path.join(ROOT, "tests", "golden", "bitstreams.json")
    stored = json.load(open(path)) if "--rehash" in sys.argv else None
    for name in STREAMS:
        aus = [bytes.fromhex(a) for a in stored[name]["access_units"]] if stored else write_stream(pkg, name)
        rec, pcm, shape = decode_stream(pkg, oracle, name, aus)
        out[name] = dict(asc=STREAMS[name][0].hex(), access_units=[a.hex() for a in aus], records_sha256=rec,
                         pcm_s16_sha256=pcm, frame_shape=shape)
        print(name, len(aus), "units,", sum(len(a) for a in aus), "bytes, pcm", pcm[:16])
    for name in LAYOUT_STREAMS:
        aus = [bytes.fromhex(a) for a in stored[name]["access_units"]] if stored and name in stored else write_layout_stream(pkg, name)
        rec, pcm, shape = decode_layout_stream(pkg, oracle, name, aus)
        out[name] = dict(asc=LAYOUT_STREAMS[name][0].hex(), access_units=[a.hex() for a in aus], records_sha256=rec,
                         pcm_s16_sha256=pcm, frame_shape=shape)
        print(name, len(aus), "units,", sum(len(a) for a in aus), "bytes, pcm", pcm[:16])
    for name in COUPLED_STREAMS:
        if stored and name in stored:
            asc, aus = bytes.fromhex(stored[name]["asc"]), [bytes.fromhex(a) for a in stored[name]["access_units"]]
        else:
            asc, aus = write_coupled_stream(pkg, oracle, name)
        rec, pcm, shape = decode_coupled_stream(pkg, oracle, name, asc, aus)
        out[name] = dict(asc=asc.hex(), access_units=[a.hex() for a in aus], records_sha256=rec, pcm_s16_sha256=pcm,
                         frame_shape=shape)
        print(name, len(aus), "units,", sum(len(a) for a in aus), "bytes, pcm", pcm[:16])
    json.dump(out, open(path, "w"), indent=0)


if __name__ == "__main__":
    main()
