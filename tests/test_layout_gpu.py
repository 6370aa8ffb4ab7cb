"""Channel layouts on the GPU: heaac_pcm_interleave_batch against ff_float_to_int16_interleave_c (dsputil.c:3989-4001,
oracle/or_core.c), and whole multi-element streams through heaac_codec_decode against the oracle decoding the same
elements one by one."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("channels,length", [(1, 1024), (2, 2048), (3, 1024), (6, 1024), (6, 2048), (8, 1024), (16, 1024)])
@pytest.mark.parametrize("sse2", [False, True])
def test_interleave_any_channel_count(pkg, oracle, dev, channels, length, sse2):
    import torch
    rng = np.random.default_rng(17 * channels + length + sse2)
    n = 5
    # planes as a layout's elements leave them: pairs in [n][2][len] buffers, single channels in [n][1][len]
    bufs, planes, host = [], [], []
    c = 0
    while c < channels:
        pair = c + 1 < channels and rng.random() < 0.6
        k = 2 if pair else 1
        if sse2:
            a = rng.standard_normal((n, k, length)).astype(np.float32) * 20000.0
            a.reshape(-1)[::97] = 70000.0; a.reshape(-1)[5::131] = -70000.0; a.reshape(-1)[3] = np.nan
        else:
            a = (385.0 + rng.standard_normal((n, k, length)) * 0.4).astype(np.float32)       # the C path's biased floats
            a.reshape(-1)[::89] = 386.5; a.reshape(-1)[7::113] = 383.0                  # beyond full scale: saturates
        t = torch.from_numpy(a).cuda()
        bufs.append(t)
        for j in range(k):
            planes.append((t, j * length, k * length))
            host.append(a[:, j])
        c += k
    got = dev.pcm_interleave(planes, length, pkg.PCM_S16_SSE2 if sse2 else pkg.PCM_S16).cpu().numpy()
    for f in range(n):
        want = oracle.float_to_int16_interleave([h[f] for h in host], sse2=sse2)
        assert np.array_equal(got[f], want), (f, np.argwhere(got[f] != want)[:4])


def test_interleave_refuses_what_it_cannot_do(pkg, dev):
    import torch
    t = torch.zeros(4096, device="cuda")
    L = pkg.lib()
    refs = (pkg._PlaneRef * 2)()
    out = torch.zeros(8192, dtype=torch.int16, device="cuda")
    refs[0].d_base = t.data_ptr(); refs[0].frame_stride = 1024
    refs[1].d_base = t.data_ptr() + 4; refs[1].frame_stride = 1024          # not 16-byte aligned
    args = lambda ch, ln, fmt: L.heaac_pcm_interleave_batch(dev._h, ch, refs, ln, fmt, C.c_void_p(out.data_ptr()), C.c_size_t(2), None)
    assert args(2, 1024, pkg.PCM_S16) == -1
    assert args(1, 1022, pkg.PCM_S16) == -1 and args(1, 1024, pkg.PCM_F32) == -1 and args(17, 1024, pkg.PCM_S16) == -1
    assert args(1, 1024, pkg.PCM_S16) == 0
