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
