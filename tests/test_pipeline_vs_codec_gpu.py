"""Two ways through the same access units: heaac_pipeline_* (batched records, csrc/pipeline.hip) and one
heaac_codec_decode context per stream (csrc/shim.hip).  On DAMAGED streams -- bit flips, byte noise, truncation,
splices from other units -- both must still say the same thing unit for unit: refused (no samples) or the same PCM,
and carry on alike afterwards.  Each path is pinned to the oracle on written damage elsewhere
(tests/test_refused_units.py); this one runs them against each other on damage nobody planned."""
import ctypes as C

import numpy as np
import pytest

import test_parse as TP
import test_pipeline as TPL
import test_sbr_parse as TS
from test_damaged_streams_gpu import _mutate

pytestmark = pytest.mark.gpu
# HEAAC_FUZZ_SEEDS=n: n seeds instead of three (a longer soak: `HEAAC_FUZZ_SEEDS=40 pytest tests/test_pipeline_vs_codec_gpu.py`)
import os
SEEDS = list(range(int(os.environ.get("HEAAC_FUZZ_SEEDS", "3"))))


@pytest.mark.parametrize("seed", SEEDS)
@pytest.mark.parametrize("mode", ["main_stereo", "lc_mono", "hev2", "hev1"])
def test_pipeline_and_codec_agree_on_damaged_streams(pkg, dev, mode, seed):
    from test_shim_gpu import HeaacCodecContext, HeaacPacket
    lib = pkg.lib()
    rng = np.random.default_rng(sum(map(ord, mode)) + 1000 * seed)
    n, ticks = 28, 11
    if mode in ("hev2", "hev1"):
        cpe = mode == "hev1"
        ch = 2 if cpe else 1
        units = TPL._ticks(pkg, rng, ch, not cpe, n, ticks)
        m4 = TS._he_cfg(pkg, ch, not cpe)
        hcfg = pkg.CFG_HEV1 if cpe else pkg.CFG_HEV2
        asc = bytes([0x2B, 0x11, 0x88, 0x00]) if cpe else bytes([0xEB, 0x09, 0x88, 0x00])
        length, nout = 2048, 2
    else:
        aot, ch = (1, 2) if mode == "main_stereo" else (2, 1)
        units = [[TP._write_au(rng, 3, aot, ch == 2, extras=False, quiet=True)[0] for _ in range(n)] for _ in range(ticks)]
        m4 = TP._cfg(pkg, aot, 3, ch)
        hcfg = pkg.CFG_LC_STEREO if ch == 2 else pkg.CFG_LC_MONO
        asc = bytes([(aot << 3) | 1, 0x80 | (ch << 3)])
        length, nout = 1024, ch
    pool = [u for tick in units for u in tick]
    fed = [list(t) for t in units]
    for t in range(2, ticks):
        for i in range(n):
            if rng.random() < 0.4:
                fed[t][i] = _mutate(rng, fed[t][i], pool)
    pl = pkg.Pipeline(m4, hcfg, n, threads=3)
    got, status = [], []
    for t in range(ticks):
        status.append(np.array(pl.submit(fed[t])).copy())
        got.append(pl.collect().copy())
    pl.close()
    codec = C.c_void_p.in_dll(lib, "heaac_aac_decoder")
    out = (C.c_int16 * (192000 // 2))()
    refused = decoded_after_refusal = 0
    for i in range(n):
        ctx = HeaacCodecContext(cfg=-1, extradata=asc, extradata_size=len(asc))
        assert lib.heaac_codec_open(C.byref(ctx), C.c_void_p(C.addressof(codec))) == 0
        seen_refusal = False
        for t in range(ticks):
            b = fed[t][i]
            buf = C.create_string_buffer(b, len(b))
            pkt = HeaacPacket(C.cast(buf, C.c_void_p), len(b))
            size = C.c_int(192000)
            used = lib.heaac_codec_decode(C.byref(ctx), out, C.byref(size), C.byref(pkt))
            if used < 0:
                # (a negative pipeline status alone does not say refused: a failed SBR payload reports one and decodes)
                assert status[t][i] < 0 and not got[t][i].any(), (mode, i, t, used, int(status[t][i]))
                refused += 1
                seen_refusal = True
                continue
            assert size.value == length * nout * 2, (mode, i, t)
            pcm = np.frombuffer(out, np.int16, length * nout).reshape(length, nout)
            assert np.array_equal(pcm, got[t][i]), (mode, i, t, int(status[t][i]))
            decoded_after_refusal += seen_refusal
        assert lib.heaac_codec_close(C.byref(ctx)) == 0
    assert refused > n // 2 and decoded_after_refusal > n
