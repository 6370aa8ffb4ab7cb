"""N > 1 path on CPU: two gloo ranks shard a frame batch by index, decode their shards
independently (the oracle stands in for the GPU kernels here -- what is tested is the
sharding and the PCM gather), rank 0 gathers and the result must equal a
single-process decode of the whole batch."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import importlib, os, sys
import numpy as np, torch, torch.distributed as dist
ROOT = sys.argv[1]; out = sys.argv[2]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g, oracle_lib as O
pkg = g.load_package()
shard = importlib.import_module("ffmpeg_heaac_amd.shard")
synth = importlib.import_module("ffmpeg_heaac_amd.synth")
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
N = 37                                   # ragged on purpose
rng = np.random.default_rng(99)          # every rank draws the same global batch
coeffs, ics = next(synth.lc_stream(rng, N, 1, 2))
state = (rng.standard_normal((N, 1024)) * 1e-3).astype(np.float32)
lo, hi = shard.shard_range(N, rank, world)
pcm, st = O.lc_decode_batch(2, coeffs[lo:hi], ics[lo:hi], state[lo:hi], O.PCM_S16)
full = shard.gather_pcm(torch.from_numpy(pcm), N, dst=0)
dist.barrier()
if rank == 0:
    ref, _ = O.lc_decode_batch(2, coeffs, ics, state, O.PCM_S16)
    assert full.shape[0] == N
    np.save(out, np.array([int(np.array_equal(full.numpy(), ref))]))
dist.destroy_process_group()
'''


def test_shard_range_covers_batch():
    sys.path.insert(0, ROOT)
    import importlib
    import __graft_entry__ as g
    g.load_package()
    shard = importlib.import_module("ffmpeg_heaac_amd.shard")
    for n in (0, 1, 7, 8, 37, 2 * 1024 * 1024):
        for w in (1, 2, 3, 8):
            r = [shard.shard_range(n, i, w) for i in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_shard_and_gather(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "ok.npy"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2",
               OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(out)], env=dict(env, RANK=str(r)))
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    assert np.load(out)[0] == 1


def _bench(*extra, env=None):
    e = dict(os.environ, OMP_NUM_THREADS="1")
    e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra], env=e,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher on the command line (the driver's shape):
    bench.py starts the ranks itself; rank 0's single JSON line comes back on stdout.
    --dry-run replaces the GPU work by a sleep, everything around it is the real code."""
    import json
    p = _bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run", "--frames", "1000",
               "--backend", "gloo", "--gather")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak" and out["dry_run"]
    assert abs(out["per_gpu_value"] * 2 - out["value"]) < 1e-6 * out["value"]
    assert out["config"]["frames_per_gpu"] == 1000 and "pcm_gather_ms" in out
    assert {"achieved", "peak", "frac", "gflops"} <= set(out["roofline"])


def test_bench_rank_failure_is_not_swallowed():
    """A rank that dies makes the launcher exit non-zero and print no JSON line."""
    p = _bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--frames", "64", "--backend", "gloo")
    # no --dry-run: without a GPU every rank fails loudly in Device()
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
