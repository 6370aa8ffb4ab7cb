"""Host tables: product (ffmpeg-heaac_amd/csrc/tables.c) vs oracle (oracle/or_core.c),
written independently from the same reference formulas -- must agree bit for bit.
Also pins a few closed-form facts about them."""
import hashlib

import numpy as np
import pytest

TABLES = ["cos16", "cos32", "cos64", "cos128", "cos256", "cos512",
          "tcos2048", "tcos256", "tcos128s", "tcos128a",
          "kbd_long", "kbd_short", "sine_long", "sine_short",
          "qmf_us", "qmf_ds", "noise", "pd_re_smooth", "pd_im_smooth", "HA", "HB",
          "f20_0_8", "f34_0_12", "f34_1_8", "f34_2_4", "Q_fract_allpass", "phi_fract",
          "revtab0", "revtab1", "revtab2"]


@pytest.mark.parametrize("name", TABLES)
def test_table_bit_identical(pkg, oracle, name):
    a = pkg.get_table(name)
    b = oracle.get_table(name)
    n = min(a.size, b.size)          # oracle cos tables also hold the mirrored half
    assert n > 0
    if name.startswith("Q_fract") or name.startswith("phi_fract"):
        # 20-band rows 30..49 are unused padding in both
        a = a.reshape(2, 50, -1).copy(); b = b.reshape(2, 50, -1).copy()
        a[0, 30:] = 0; b[0, 30:] = 0
        a = a.ravel(); b = b.ravel()
    assert np.array_equal(a[:n].view(np.uint32), b[:n].view(np.uint32)), name


def test_sqrthalf_is_cos16_2(pkg):
    # fft16 (fft.c:327-339) uses the literal sqrthalf where the generic pass would
    # read ff_cos_16[2]; the kernels rely on the two being the same float.
    assert pkg.get_table("cos16")[2] == np.float32(0.70710678118654752440)


def test_qmf_window_symmetry(pkg):
    us = pkg.get_table("qmf_us")
    ds = pkg.get_table("qmf_ds")
    assert np.array_equal(ds, us[::2])
    idx = np.arange(1, 320)
    sign = np.ones(319, np.float32)
    sign[idx == 64] = -1      # tap 384
    sign[idx == 192] = -1     # tap 512
    assert np.array_equal(us[320 + idx], sign * us[320 - idx])


def test_windows_power_complementary(pkg):
    # Princen-Bradley: w[i]^2 + w[N-1-i]^2 == 1
    for name in ("kbd_long", "kbd_short", "sine_long", "sine_short"):
        w = pkg.get_table(name).astype(np.float64)
        assert np.abs(w ** 2 + w[::-1] ** 2 - 1).max() < 1e-6, name


def test_revtab_is_permutation(pkg):
    for name, n in (("revtab0", 512), ("revtab1", 64), ("revtab2", 32)):
        r = pkg.get_table(name).astype(int)
        assert sorted(r) == list(range(n))


def test_table_checksums_recorded(pkg):
    """Checksums of every table as built in THIS environment; compared with the
    committed fixture so a libm difference on another box is caught, not absorbed."""
    import json, os
    path = os.path.join(os.path.dirname(__file__), "golden", "table_sha256.json")
    got = {n: hashlib.sha256(pkg.get_table(n).tobytes()).hexdigest() for n in TABLES}
    if not os.path.exists(path):
        pytest.skip("fixture missing; run tests/golden/make_table_checksums.py")
    want = json.load(open(path))
    assert got == want
