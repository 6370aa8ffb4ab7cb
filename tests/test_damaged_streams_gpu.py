"""Damaged bitstreams end to end: access units of the stored golden streams, mutated (bit flips, byte noise,
truncation, splices), go through the host parser on streams that keep their state; whatever records come out
(all of them pass validate.h, tests/test_parse_fuzz.py) are decoded on the GPU and by the oracle.  The damaged
frames reach record combinations no writer in this suite produces on purpose; the two paths must still agree bit
for bit wherever the oracle's output is finite."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mutate(rng, au, pool):
    b = bytearray(au)
    mode = int(rng.integers(0, 5))
    if mode == 0:
        for _ in range(int(rng.integers(1, 9))):
            b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
    elif mode == 1:
        for _ in range(int(rng.integers(1, 7))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
    elif mode == 2 and len(b) > 8:
        b = b[: int(rng.integers(4, len(b)))] + bytearray(8)
    elif mode == 3:
        other = pool[int(rng.integers(0, len(pool)))]
        at = int(rng.integers(0, len(b)))
        for i in range(at, len(b)):
            b[i] = other[i % len(other)]
    return bytes(b)


@pytest.mark.parametrize("name", ["hev2_mono_24k", "hev1_stereo_24k"])
def test_damaged_streams_decode_like_the_oracle(pkg, oracle, dev, name):
    import torch
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_bitstream_vectors as B
    v = json.load(open(os.path.join(ROOT, "tests", "golden", "bitstreams.json")))[name]
    seeds = [bytes.fromhex(a) for a in v["access_units"]]
    asc, si, cpe, sbr, ps, frames, seed = B.STREAMS[name]
    ch = 2 if cpe else 1
    hcfg = pkg.CFG_HEV1 if cpe else pkg.CFG_HEV2
    m4, _ = pkg.asc_parse(asc)
    m4.sbr = 1
    if ps:
        m4.ps = 1
    rng = np.random.default_rng(4242 + cpe)
    n, steps = 96, 14
    tab = pkg.SbrHeaderTable(256)
    st, sst = np.zeros(n, pkg.AAC_STREAM_DT), pkg.sbr_streams(n)
    state = np.zeros((n, pkg.STATE_WORDS[hcfg]), np.float32)
    rs = np.full(n, 0x1f2e3d4c, np.int32)
    d_rs = torch.from_numpy(rs.copy()).cuda()
    compared = damaged_started = 0
    for step in range(steps):
        aus = []
        for s in range(n):
            au = seeds[(step + s) % len(seeds)]
            aus.append(_mutate(rng, au, seeds) if rng.random() < 0.5 else au)
        p = pkg.heaac_parse_batch(m4, st, sst, tab, aus, with_ps=ps)
        good = p["info"]["channels"] == ch                   # the core element parsed: the frame is decodable
        # frames whose core element was refused carry no spectrum: silence with a start = 0 record on both sides
        p["coeffs"][~good] = 0
        p["tools"][~good] = np.zeros(1, pkg.TOOLS_FRAME_DT)
        p["ics"][~good] = np.zeros(1, pkg.ICS_DT)
        p["sbr"]["start"][~good] = 0
        if ps:
            p["ps"]["start"][~good] = 0
        hdr = tab.headers()
        for s in range(n):
            assert pkg.validate_frame(hcfg, p["sbr"][s:s + 1], hdr, p["ps"][s:s + 1] if ps else None) == "NONE", (step, s)
        coeffs = np.ascontiguousarray(p["coeffs"][:, :ch])
        ics = np.ascontiguousarray(p["ics"][:, :ch])
        ref_c, rs = oracle.spectral_tools_batch(ch, coeffs, p["tools"], rng=rs)
        d_c = torch.from_numpy(coeffs).cuda()
        dev.spectral_tools(ch, d_c, pkg.to_device(p["tools"]), rng=d_rs)
        same = d_c.cpu().numpy().view(np.uint32) == ref_c.view(np.uint32)
        finite_c = np.isfinite(ref_c).all(axis=(1, 2))
        assert same[finite_c].all(), step
        # damaged scalefactors reach 1e30: bring every stream's spectrum to audio level by a power of two, and
        # silence the ones the spectral tools drove to infinity
        top = np.abs(np.where(np.isfinite(ref_c), ref_c, 0)).max(axis=(1, 2))
        scale = (2.0 ** -np.ceil(np.log2(np.maximum(top, 1e-3) / 1e-3))).astype(np.float32)
        scale[~finite_c] = 0
        ref_c = np.where(finite_c[:, None, None], ref_c, 0) * scale[:, None, None]
        ref_c = ref_c.astype(np.float32)
        d_c = torch.from_numpy(ref_c).cuda()
        state_in = state
        ref_pcm, state = oracle.he_decode_batch(hcfg, ref_c, ics, p["sbr"], hdr, p["ps"] if ps else None, state_in, pkg.PCM_F32)
        pcm, d_state = dev.he_decode(hcfg, d_c, pkg.to_device(ics), pkg.to_device(p["sbr"]), pkg.to_device(hdr),
                                     pkg.to_device(p["ps"]) if ps else None, torch.from_numpy(state_in).cuda())
        fin = np.isfinite(ref_pcm).all(axis=(1, 2)) & np.isfinite(state).all(axis=1)
        got, gst = pcm.cpu().numpy(), d_state.cpu().numpy()
        bad = (got.view(np.uint32) != ref_pcm.view(np.uint32)).any(axis=(1, 2)) | (gst.view(np.uint32) != state.view(np.uint32)).any(axis=1)
        assert not (bad & fin).any(), (step, np.nonzero(bad & fin)[0][:8].tolist())
        compared += int(fin.sum())
        damaged_started += int(((p["status"] != 0) & (p["sbr"]["start"] == 1)).sum())
        # a stream that went non-finite starts over (both sides from the same zero state)
        state[~fin] = 0
    assert compared > 0.9 * n * steps
