"""include/heaac_multi.h: the C-level multi-device entry.  CPU: the partition and the loud failure without a
device.  GPU (one card): two device slots on the same card decode two shards concurrently from two host threads and
gather their PCM -- the result equals the single-context decode of the whole batch bit for bit."""
import ctypes as C
import importlib

import numpy as np
import pytest


class HeShard(C.Structure):
    _fields_ = [("d_coeffs", C.c_void_p), ("d_ics", C.c_void_p), ("d_sbr", C.c_void_p), ("d_hdr", C.c_void_p),
                ("n_hdr", C.c_size_t), ("d_ps", C.c_void_p), ("d_state_in", C.c_void_p), ("d_state_out", C.c_void_p),
                ("d_pcm", C.c_void_p), ("n", C.c_size_t)]


def test_partition_is_the_python_harness_partition(pkg):
    shard = importlib.import_module("ffmpeg_heaac_amd.shard")
    f = pkg.lib().heaac_multi_shard
    f.restype = None
    for n in (0, 1, 7, 8, 9, 1000, 262144, 2 * 1024 * 1024 + 5):
        for G in (1, 2, 3, 4, 8, 16):
            covered = 0
            for g in range(G):
                first, count = C.c_size_t(), C.c_size_t()
                f(C.c_size_t(n), g, G, C.byref(first), C.byref(count))
                lo, hi = shard.shard_range(n, g, G)
                assert (first.value, count.value) == (lo, hi - lo), (n, G, g)
                assert first.value == covered
                covered += count.value
            assert covered == n
    first, count = C.c_size_t(5), C.c_size_t(5)
    f(C.c_size_t(10), 3, 2, C.byref(first), C.byref(count))            # a slot outside the set owns nothing
    assert (first.value, count.value) == (0, 0)


def test_create_fails_loudly_without_a_device_and_on_bad_arguments(pkg):
    import torch
    lib = pkg.lib()
    m = C.c_void_p(1)
    devs = (C.c_int * 2)(0, 0)
    assert lib.heaac_multi_create(C.byref(m), devs, 0, C.c_size_t(64)) == -1 and not m.value          # HEAAC_ERR_ARG
    assert lib.heaac_multi_create(C.byref(m), devs, 17, C.c_size_t(64)) == -1
    if not torch.cuda.is_available():
        assert lib.heaac_multi_create(C.byref(m), devs, 2, C.c_size_t(64)) == -4 and not m.value      # HEAAC_ERR_NODEVICE
    assert lib.heaac_multi_devices(None) == 0


@pytest.mark.gpu
def test_two_slots_decode_and_gather_like_one_context(pkg, oracle, dev):
    import torch
    synth = importlib.import_module("ffmpeg_heaac_amd.synth")
    lib = pkg.lib()
    lib.heaac_multi_stream.restype = C.c_void_p
    rng = np.random.default_rng(12)
    cfg, n = pkg.CFG_HEV2, 301                                          # an odd count: shards of 151 and 150
    hdr = synth.default_headers(pkg)
    frames = list(synth.he_stream(rng, cfg, n, 2, hdr))
    words = pkg.STATE_WORDS[cfg]
    d_hdr = pkg.to_device(hdr)
    m = C.c_void_p()
    devs = (C.c_int * 2)(0, 0)
    assert lib.heaac_multi_create(C.byref(m), devs, 2, C.c_size_t(256)) == 0
    assert lib.heaac_multi_devices(m) == 2
    st_multi = [torch.zeros((n, words), device="cuda")]
    st_one = torch.zeros((n, words), device="cuda")
    state = np.zeros((n, words), np.float32)
    for fr in frames:
        t = dict(coeffs=torch.from_numpy(fr["coeffs"]).cuda(), ics=pkg.to_device(fr["ics"]), sbr=pkg.to_device(fr["sbr"]),
                 ps=pkg.to_device(fr["ps"]))
        pcm_one, st_one = dev.he_decode(cfg, t["coeffs"], t["ics"], t["sbr"], d_hdr, t["ps"], st_one, pcm_format=pkg.PCM_S16)
        torch.cuda.synchronize()
        # per-shard views of the same arrays (every array is frame-major, so a shard is a slice) and per-shard PCM
        gathered = torch.zeros((n, 2048, 2), dtype=torch.int16, device="cuda")
        shard_pcm, shards = [], (HeShard * 2)()
        for g in range(2):
            first, count = C.c_size_t(), C.c_size_t()
            lib.heaac_multi_shard(C.c_size_t(n), g, 2, C.byref(first), C.byref(count))
            lo, cnt = first.value, count.value
            shard_pcm.append(torch.zeros((cnt, 2048, 2), dtype=torch.int16, device="cuda"))
            s = shards[g]
            s.d_coeffs = t["coeffs"].data_ptr() + lo * 4096
            s.d_ics = t["ics"].data_ptr() + lo * pkg.ICS_DT.itemsize
            s.d_sbr = t["sbr"].data_ptr() + lo * pkg.SBR_FRAME_DT.itemsize
            s.d_hdr, s.n_hdr = d_hdr.data_ptr(), len(hdr)
            s.d_ps = t["ps"].data_ptr() + lo * pkg.PS_FRAME_DT.itemsize
            s.d_state_in = s.d_state_out = st_multi[0].data_ptr() + lo * words * 4
            s.d_pcm, s.n = shard_pcm[g].data_ptr(), cnt
        torch.cuda.synchronize()
        assert lib.heaac_multi_he_decode(m, cfg, 0, shards, pkg.PCM_S16, C.c_void_p(gathered.data_ptr()), 0) == 0
        assert torch.equal(gathered, pcm_one)
        assert torch.equal(torch.cat(shard_pcm), pcm_one)
        assert torch.equal(st_multi[0], st_one)
        ref, state = oracle.he_decode_batch(cfg, fr["coeffs"], fr["ics"], fr["sbr"], hdr, fr["ps"], state, oracle.PCM_S16)
        assert np.array_equal(gathered.cpu().numpy(), ref)
    # a shard with a bad argument fails the call, the others still complete
    shards[1].d_sbr = None
    assert lib.heaac_multi_he_decode(m, cfg, 0, shards, pkg.PCM_S16, None, 0) == -1
    lib.heaac_multi_destroy(m)
