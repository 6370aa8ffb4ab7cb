"""Golden vectors: SHA-256 of PCM + state for seeded synthetic streams (tests/golden/pcm_sha256.json,
minted by tests/golden/make_pcm_checksums.py).  The CPU test pins the oracle against drift; the GPU
test checks the HIP path against the same hashes WITHOUT calling the oracle."""
import importlib, json, os, sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_pcm_checksums as G  # noqa: E402

PATH = os.path.join(os.path.dirname(__file__), "golden", "pcm_sha256.json")


def _synth():
    import __graft_entry__ as g
    return importlib.import_module(g.PKG_NAME + ".synth")


@pytest.mark.parametrize("name", sorted(G.CASES))
def test_oracle_reproduces_golden(pkg, oracle, name):
    want = json.load(open(PATH))
    got = G.run_case(name, pkg, _synth(), oracle.lc_decode_batch, oracle.he_decode_batch, oracle.spectral_tools_batch)
    assert got == want[name]


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(G.CASES))
def test_hip_path_reproduces_golden(pkg, dev, name):
    import torch

    def lc(ch, coeffs, ics, state, fmt):
        pcm, st = dev.lc_decode(ch, torch.from_numpy(coeffs).cuda(), pkg.to_device(ics),
                                torch.from_numpy(np.ascontiguousarray(state)).cuda(), pcm_format=fmt)
        return pcm.cpu().numpy(), st.cpu().numpy()

    def he(cfg, coeffs, ics, sbr, hdr, ps, state, fmt):
        pcm, st = dev.he_decode(cfg, torch.from_numpy(coeffs).cuda(), pkg.to_device(ics), pkg.to_device(sbr),
                                pkg.to_device(hdr), pkg.to_device(ps) if ps is not None else None,
                                torch.from_numpy(np.ascontiguousarray(state)).cuda(), pcm_format=fmt)
        return pcm.cpu().numpy(), st.cpu().numpy()

    def tools(ch, c, t, rs):
        d = torch.from_numpy(c).cuda(); r = torch.from_numpy(rs.copy()).cuda()
        dev.spectral_tools(ch, d, pkg.to_device(t), rng=r)
        return d.cpu().numpy(), r.cpu().numpy()

    want = json.load(open(PATH))
    assert G.run_case(name, pkg, _synth(), lc, he, tools) == want[name]


# ---------------------------------------------------------------------------------------------
# bitstream vectors: stored access units -> records and int16 PCM (tests/golden/bitstreams.json)
# ---------------------------------------------------------------------------------------------
import make_bitstream_vectors as B  # noqa: E402

BPATH = os.path.join(os.path.dirname(__file__), "golden", "bitstreams.json")


@pytest.mark.parametrize("name", sorted(B.STREAMS))
def test_stored_bitstreams_parse_and_decode_to_the_stored_hashes(pkg, oracle, name):
    """The committed access units go through the host parser and the oracle: per-frame record hashes and the
    PCM hash are the stored ones; and the writers still produce these very bytes from their seeds."""
    v = json.load(open(BPATH))[name]
    aus = [bytes.fromhex(a) for a in v["access_units"]]
    rec, pcm, shape = B.decode_stream(pkg, oracle, name, aus)
    assert rec == v["records_sha256"] and pcm == v["pcm_s16_sha256"] and shape == v["frame_shape"]
    assert [a.hex() for a in B.write_stream(pkg, name)] == v["access_units"]


@pytest.mark.parametrize("name", sorted(B.LAYOUT_STREAMS))
def test_stored_multichannel_bitstreams_parse_and_decode_to_the_stored_hashes(pkg, oracle, name):
    """The same for the stored 5.1 streams: layout parser + the oracle per element + the reference's interleave."""
    v = json.load(open(BPATH))[name]
    aus = [bytes.fromhex(a) for a in v["access_units"]]
    rec, pcm, shape = B.decode_layout_stream(pkg, oracle, name, aus)
    assert rec == v["records_sha256"] and pcm == v["pcm_s16_sha256"] and shape == v["frame_shape"]
    assert [a.hex() for a in B.write_layout_stream(pkg, name)] == v["access_units"]


@pytest.mark.parametrize("name", sorted(B.COUPLED_STREAMS))
def test_stored_coupled_bitstreams_parse_and_decode_to_the_stored_hashes(pkg, oracle, name):
    """The same for the stored streams with coupling channel elements (a program config element names them)."""
    v = json.load(open(BPATH))[name]
    asc, aus = bytes.fromhex(v["asc"]), [bytes.fromhex(a) for a in v["access_units"]]
    rec, pcm, shape = B.decode_coupled_stream(pkg, oracle, name, asc, aus)
    assert rec == v["records_sha256"] and pcm == v["pcm_s16_sha256"] and shape == v["frame_shape"]
    asc2, aus2 = B.write_coupled_stream(pkg, oracle, name)
    assert asc2.hex() == v["asc"] and [a.hex() for a in aus2] == v["access_units"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(B.STREAMS) + sorted(B.LAYOUT_STREAMS) + sorted(B.COUPLED_STREAMS))
def test_codec_decodes_stored_bitstreams_to_the_stored_pcm(pkg, name):
    """heaac_codec_open / _decode on the committed bytes (cfg from the stream): the int16 PCM hashes to the
    stored value.  No oracle in this test."""
    import ctypes as C
    import hashlib
    from test_shim_gpu import HeaacCodecContext, HeaacPacket
    lib = pkg.lib()
    v = json.load(open(BPATH))[name]
    asc = bytes.fromhex(v["asc"])
    ctx = HeaacCodecContext(cfg=-1, extradata=asc, extradata_size=len(asc))
    codec = C.c_void_p.in_dll(lib, "heaac_aac_decoder")
    assert lib.heaac_codec_open(C.byref(ctx), C.c_void_p(C.addressof(codec))) == 0
    out = (C.c_int16 * (192000 // 2))()
    m = hashlib.sha256()
    for a in v["access_units"]:
        au = bytes.fromhex(a)
        buf = C.create_string_buffer(au, len(au))
        pkt = HeaacPacket(C.cast(buf, C.c_void_p), len(au))
        size = C.c_int(192000)
        assert lib.heaac_codec_decode(C.byref(ctx), out, C.byref(size), C.byref(pkt)) == len(au)
        assert size.value == 2 * v["frame_shape"][0] * v["frame_shape"][1]
        m.update(memoryview(out).cast("B")[: size.value])
    assert m.hexdigest() == v["pcm_s16_sha256"]
    assert lib.heaac_codec_close(C.byref(ctx)) == 0
