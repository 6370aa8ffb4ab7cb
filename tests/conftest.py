import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package; built on demand (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g
    p = g.load_package()
    if not os.path.exists(p.LIB_PATH):
        g.build()
    return p


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def dev(pkg):
    """HeaacDevice on cuda:0.  GPU tests only; fails loudly without the HIP library."""
    import torch
    assert torch.cuda.is_available(), "gpu-marked test started without a GPU"
    torch.cuda.set_device(0)
    d = pkg.Device()
    yield d
    d.close()
