"""The twelve-wave HF + PS kernel (k_hfps12, profiles/r04_experiments.md E1) is not shipped -- it is slower -- but it is
kept buildable (`tools/build_variants.sh tuning "-DHEAAC_TUNING"` -> ab/libtuning.so, launched with HEAAC_HFPS12=1), and
what is kept is kept correct: the HE-AACv2 parity tests run against it in a child process.  Skipped when the variant
library has not been built (it is a measurement artefact, not part of build())."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "ab", "libtuning.so")


@pytest.mark.gpu
def test_twelve_wave_variant_is_bit_exact():
    if not os.path.exists(LIB):
        pytest.skip("ab/libtuning.so not built (tools/build_variants.sh tuning \"-DHEAAC_TUNING\")")
    import glob
    srcs = glob.glob(os.path.join(ROOT, "ffmpeg-heaac_amd", "csrc", "*.[ch]*")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    if any(os.path.getmtime(f) > os.path.getmtime(LIB) for f in srcs if not f.endswith((".o", ".so"))):
        pytest.skip("ab/libtuning.so is older than the sources: rebuild it to run this test")
    env = dict(os.environ, HEAAC_LIB_PATH=LIB, HEAAC_HFPS12="1")
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_he_gpu.py"),
                        os.path.join(ROOT, "tests", "test_golden.py"), "-m", "gpu", "-x", "-q", "-k", "hev2 or unstored or golden or codec"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200)
    tail = "\n".join(p.stdout.splitlines()[-15:])
    assert p.returncode == 0, tail
    assert " passed" in tail, tail
    # (and the child really ran on the variant library)
    q = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import __graft_entry__ as g; "
                        "p = g.load_package(); print(p.LIB_OVERRIDDEN, p.LIB_PATH)" % ROOT],
                       env=env, stdout=subprocess.PIPE, text=True, timeout=300)
    assert q.stdout.split()[:2] == ["True", LIB], q.stdout
