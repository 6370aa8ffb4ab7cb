"""The C ABI from a plain C program (tests/c/abi_check.c, built with gcc against include/*.h and linked to
the product library only): struct layouts against `struct AVCodec`, host entry points, loud failure without
a device; on the GPU the transform plugin surface by the reference's own fft-test criterion and the
AVCodec-shaped decoder."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "ffmpeg-heaac_amd")
EXE = os.path.join(ROOT, "tests", "c", "_build", "abi_check")


def _build(pkg):
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    src = os.path.join(ROOT, "tests", "c", "abi_check.c")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(src), os.path.getmtime(pkg.LIB_PATH)):
        subprocess.check_call(["gcc", "-std=gnu99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src,
                               "-o", EXE, "-L", LIBDIR, "-l:libheaac_amd.so", "-Wl,-rpath," + LIBDIR, "-lm"])
    return EXE


def _run(mode):
    p = subprocess.run([EXE, mode], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0, p.stdout
    assert p.stdout.strip().endswith("ok"), p.stdout
    return p.stdout


def test_c_program_layouts_and_host_entry_points(pkg):
    import torch
    _build(pkg)
    _run("cpu" if torch.cuda.is_available() else "nodevice")


@pytest.mark.gpu
def test_c_program_transforms_and_codec_on_gpu(pkg):
    _build(pkg)
    out = _run("gpu")
    assert "imdct N=2048" in out and "fft 512" in out
