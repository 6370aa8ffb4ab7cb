#!/usr/bin/env python3
"""bench.py -- HE-AAC decode DSP throughput on MI355X (driver contract).

  python bench.py --gpus N --steps K --warmup W [--workload hev2|hev1|lc_stereo]

A "step" is one pass of the hot path over one batch of synthetic frames, inputs
resident in HBM, state chained from step to step.  N > 1 is launched by
torch.distributed.run, one rank per GPU; the batch is per-GPU (weak scaling), no
data-path collective (frames are independent).

Prints ONE JSON line on rank 0 with `roofline` (algorithmic bytes / kernel time,
HIP-event timed on the launch stream) and `cpu_baseline` (the oracle, timed on the
host on a bounded sample of the same workload; rank 0, N = 1 only).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {
    # name: (cfg attr, frames per GPU, BASELINE.json config)
    "lc_stereo": ("CFG_LC_STEREO", 64 * 1024, "AAC-LC stereo 48 kHz, 64 k-frame batch"),
    "hev1": ("CFG_HEV1", 64 * 1024, "HE-AACv1 stereo 48 kHz, 64 k-frame batch"),
    "hev2": ("CFG_HEV2", 256 * 1024, "HE-AACv2 stereo 48 kHz, 256 k-frame batch"),
    # secondary run of SURVEY s8d: 34-band PS with IPD/OPD (the general PS kernel finishes every frame)
    "hev2_34": ("CFG_HEV2", 64 * 1024, "HE-AACv2 stereo 48 kHz, 34-band PS + IPD/OPD, 64 k-frame batch"),
}
PS_MODE = {"hev2_34": "34"}
ALGO_BYTES_OVERRIDE = {"hev2_34": 95788}       # PS state 17 996 B instead of 12 744 B
# Arithmetic per frame (f32 adds + multiplies of the reference's dataflow, counted per stage in
# DESIGN.md s5; SURVEY s8d quotes ~0.6 MFLOP for HE-AACv2).
ALGO_FLOPS = {"lc_stereo": 57e3, "hev1": 0.50e6, "hev2": 0.60e6, "hev2_34": 0.68e6}
# 256 CUs x 4 SIMD-32 x 2.4 GHz, one add or multiply per lane and cycle (no FMA: -ffp-contract=off)
VALU_NOFMA_PEAK_GFLOPS = 256 * 4 * 32 * 2.4


def make_inputs(pkg, synth, torch, cfg, n, seed, pool=4096, ps_mode="20"):
    """Synthetic per-frame inputs on the GPU.  Parameters for `pool` independent
    streams are generated on the host for 3 consecutive frames (2 warm-up frames
    build a realistic state, the third is the timed one) and tiled to n frames;
    coefficients are drawn per frame on the device."""
    import numpy as np
    rng = np.random.default_rng(seed)
    pool = min(pool, n)
    reps = (n + pool - 1) // pool
    steps = []
    if cfg == pkg.CFG_LC_STEREO:
        gen = synth.lc_stream(rng, pool, 3, channels=2)
        for coeffs, ics in gen:
            steps.append(dict(ics=pkg.to_device(ics).repeat(reps)[: n * 2 * 4].contiguous()))
        hdr = None
    else:
        hdr = synth.default_headers(pkg)
        for fr in synth.he_stream(rng, cfg, pool, 3, hdr, ps_mode=ps_mode):
            d = dict(ics=pkg.to_device(fr["ics"]).repeat(reps)[: n * pkg.CORE_CH[cfg] * 4].contiguous(),
                     sbr=pkg.to_device(fr["sbr"]).repeat(reps)[: n * 680].contiguous(),
                     ps=None)
            if fr["ps"] is not None:
                d["ps"] = pkg.to_device(fr["ps"]).repeat(reps)[: n * 532].contiguous()
            d["ws_short"] = torch.from_numpy(
                (fr["ics"]["window_sequence"][:, :, 0] == 2)).cuda().repeat(reps, 1)[:n]
            steps.append(d)
        hdr = pkg.to_device(hdr)
    # coefficients: uniform +-4096*|sf_scale|, band-limited for the 24 kHz HE core
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    ch = pkg.CORE_CH[cfg]
    amp = 4096.0 / (1024.0 * 32768.0)
    coeffs = []
    for s in range(3):
        c = (torch.rand((n, ch, 1024), generator=g, device="cuda", dtype=torch.float32) * 2 - 1) * amp
        if cfg != pkg.CFG_LC_STEREO:
            k = torch.arange(1024, device="cuda")
            long_mask = (k < 400).float()
            short_mask = ((k % 128) < 50).float()
            m = torch.where(steps[s]["ws_short"][:, :, None], short_mask, long_mask)
            c = c * m
        coeffs.append(c.contiguous())
    return steps, coeffs, hdr


def run_step(pkg, dev, cfg, step, coeffs, hdr, st_in, st_out, pcm, fmt):
    if cfg == pkg.CFG_LC_STEREO:
        dev.lc_decode(2, coeffs, step["ics"], st_in, state_out=st_out, pcm=pcm, pcm_format=fmt)
    else:
        dev.he_decode(cfg, coeffs, step["ics"], step["sbr"], hdr, step["ps"], st_in,
                      state_out=st_out, pcm=pcm, pcm_format=fmt)


def cpu_baseline(pkg, synth, cfg, seconds=10.0):
    """Time the oracle (scalar C port of the reference path) on the host: 1 thread, then one thread
    per core the process may use (threads over disjoint frame ranges; the reference has no
    intra-stream threading).  ctypes releases the GIL during the C call."""
    import threading
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    rng = np.random.default_rng(5)
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    per = 512 if cfg == pkg.CFG_LC_STEREO else 64          # frames per thread and call
    n = per * ncores
    if cfg == pkg.CFG_LC_STEREO:
        frames = list(synth.lc_stream(rng, n, 3, channels=2))
        state = np.zeros((n, 1024), np.float32)
        sl = lambda fr, a, b: (fr[0][a:b], fr[1][a * 2:b * 2])
        run = lambda fr, st: O.lc_decode_batch(2, fr[0], fr[1], st, O.PCM_F32)
    else:
        hdr = synth.default_headers(pkg)
        frames = list(synth.he_stream(rng, cfg, n, 3, hdr))
        state = np.zeros((n, pkg.STATE_WORDS[cfg]), np.float32)
        nc = pkg.CORE_CH[cfg]
        sl = lambda fr, a, b: dict(coeffs=fr["coeffs"][a:b], ics=fr["ics"][a * nc:b * nc], sbr=fr["sbr"][a:b],
                                   ps=None if fr["ps"] is None else fr["ps"][a:b])
        run = lambda fr, st: O.he_decode_batch(cfg, fr["coeffs"], fr["ics"], fr["sbr"], hdr, fr["ps"], st,
                                               O.PCM_F32)
    for fr in frames[:2]:
        _, state = run(fr, state)

    def timed(nthreads, budget):
        parts = [(sl(frames[2], t * per, (t + 1) * per), np.ascontiguousarray(state[t * per:(t + 1) * per]))
                 for t in range(nthreads)]
        counts = [0] * nthreads
        t0 = time.perf_counter()

        def work(t):
            while time.perf_counter() - t0 < budget:
                run(*parts[t])
                counts[t] += per
        th = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        dt = time.perf_counter() - t0
        return sum(counts) / dt, sum(counts), dt

    v1, c1, d1 = timed(1, seconds / 2)
    vn, cn, dn = timed(ncores, seconds / 2)
    return dict(value=v1, unit="frames/s", cores=1, kind="port",
                all_cores=dict(value=vn, cores=ncores, nproc=os.cpu_count()),
                sample="oracle C port (gcc -O3 -fno-tree-vectorize -ffp-contract=off) on the timed frame of a "
                       "%d-frame synthetic set, state two frames warm: %d frames in %.1f s on 1 thread; %d frames "
                       "in %.1f s on %d threads (disjoint frame ranges)" % (n, c1, d1, cn, dn, ncores))


def self_launch(n):
    """Start `n` ranks of this script under torch.distributed.run (one per GPU, rendezvous on
    127.0.0.1), relay rank 0's JSON line, return non-zero if any rank failed."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if p.returncode != 0 or line is None:
        print("bench.py: %d-rank launch failed (exit %d)" % (n, p.returncode), file=sys.stderr)
        return p.returncode or 1
    print(line)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=None, help="frames per GPU (default: BASELINE config)")
    ap.add_argument("--pcm", default="f32", choices=["f32", "s16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo to rehearse)")
    ap.add_argument("--gather", action="store_true",
                    help="also time the PCM gather onto rank 0 (BASELINE config 5), reported separately")
    ap.add_argument("--gather-check", action="store_true",
                    help="with --gather: SHA-256 of every rank's PCM shard and of what rank 0 gathered (tests)")
    ap.add_argument("--dry-run", action="store_true",
                    help="rehearse launch / rendezvous / barriers / JSON without touching a GPU (CPU tests)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as the driver types it: start the N ranks ourselves, BEFORE
        # torch or the GPU is touched (a process that has initialised the GPU must not re-exec).
        sys.exit(self_launch(args.gpus))

    import torch
    import __graft_entry__ as g
    pkg = g.load_package()
    synth = importlib.import_module(g.PKG_NAME + ".synth")

    workload = args.workload or "hev2"
    cfg_name, frames, cfg_desc = WORKLOADS[workload]
    cfg = getattr(pkg, cfg_name)
    n = args.frames or frames
    dry = args.dry_run

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        if os.environ.get("HEAAC_BENCH_SINGLE_DEVICE"):      # rehearsal: several ranks on one card
            local = 0
        if args.backend == "nccl" and not dry:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo" if dry else args.backend)
    if world != args.gpus:
        sys.exit("bench.py: WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    on_gpu_comm = dist is not None and args.backend == "nccl" and not dry

    def sync():
        if not dry:
            torch.cuda.synchronize()

    if dry:
        # no GPU: every rank "processes" its shard of the index range (shard.py) in a timed sleep, so
        # the launcher, the rendezvous, the barriers, the max-over-ranks clock and the JSON are real
        shard = importlib.import_module(g.PKG_NAME + ".shard")
        lo, hi = shard.shard_range(n * world, rank, world)
        assert hi - lo == n

        def one(i):
            time.sleep(0.002)
        fmt = pkg.PCM_F32 if args.pcm == "f32" else pkg.PCM_S16
        pcm = torch.zeros((min(n, 64), pkg.OUT_CH[cfg], pkg.OUT_LEN[cfg]))
    else:
        torch.cuda.set_device(local)
        dev = pkg.Device(n)
        steps_in, coeffs, hdr = make_inputs(pkg, synth, torch, cfg, n, seed=1234 + rank,
                                            ps_mode=PS_MODE.get(workload, "20"))
        fmt = pkg.PCM_F32 if args.pcm == "f32" else pkg.PCM_S16
        words = pkg.STATE_WORDS[cfg]
        # state is updated in place (st_in == st_out), as a decoder does frame after frame
        st = [torch.zeros((n, words), device="cuda")] * 2
        if fmt == pkg.PCM_F32:
            pcm = torch.empty((n, pkg.OUT_CH[cfg], pkg.OUT_LEN[cfg]), device="cuda")
        else:
            pcm = torch.empty((n, pkg.OUT_LEN[cfg], pkg.OUT_CH[cfg]), dtype=torch.int16, device="cuda")

        # two state-building frames (not warm-up steps: they make state_in realistic)
        run_step(pkg, dev, cfg, steps_in[0], coeffs[0], hdr, st[0], st[1], pcm, fmt)
        run_step(pkg, dev, cfg, steps_in[1], coeffs[1], hdr, st[1], st[0], pcm, fmt)
        torch.cuda.synchronize()

        def one(i):
            # the timed frame: same parameters every step, state ping-pongs
            run_step(pkg, dev, cfg, steps_in[2], coeffs[2], hdr, st[i & 1], st[(i + 1) & 1], pcm, fmt)

    for i in range(args.warmup):
        one(i)
    sync()
    if dist is not None:
        dist.barrier()
    sync()

    ev = []
    if not dry:
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        if ev:
            ev[i][0].record()
        one(i)
        if ev:
            ev[i][1].record()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0

    # device time of one step's launches: HIP events on the launch stream (the library launches on
    # torch's current stream, the one these events are recorded on)
    kern_ms = (sum(a.elapsed_time(b) for a, b in ev) if ev else elapsed * 1e3) / max(1, args.steps)
    gather_ms, gather_check = None, None
    if args.gather and dist is not None:
        shard = importlib.import_module(g.PKG_NAME + ".shard")
        src = pcm if on_gpu_comm else pcm.cpu()
        sync(); dist.barrier(); tg = time.perf_counter()
        full = shard.gather_pcm(src, src.shape[0] * world, dst=0)
        sync(); dist.barrier()
        gather_ms = (time.perf_counter() - tg) * 1e3
        if args.gather_check:
            import hashlib
            mine = hashlib.sha256(pcm.cpu().numpy().tobytes()).hexdigest()
            hashes = [None] * world
            dist.all_gather_object(hashes, mine)
            gather_check = dict(shards=hashes, device=[local, torch.cuda.current_device() if not dry else None],
                                gathered=hashlib.sha256(full.cpu().numpy().tobytes()).hexdigest() if rank == 0 else None)
    if dist is not None:
        t = torch.tensor([elapsed, kern_ms], device="cuda" if on_gpu_comm else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kern_ms = float(t[0]), float(t[1])

    if rank == 0:
        total_frames = n * world * args.steps
        value = total_frames / elapsed
        bytes_per_frame = ALGO_BYTES_OVERRIDE.get(workload, pkg.ALGO_BYTES[cfg])
        if fmt == pkg.PCM_S16:      # int16 PCM out instead of f32: 2 bytes/sample less
            bytes_per_frame -= pkg.OUT_CH[cfg] * pkg.OUT_LEN[cfg] * 2
        achieved = bytes_per_frame * n / (kern_ms * 1e-3) / 1e9
        # HBM bytes per step: NOT measured in this run (the PMC passes need rocprofv3 around the process) but read
        # from profiles/traffic.json, where tools/traffic.sh left it for this workload and PCM format (rocprofv3
        # --pmc FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 correction applied).  An entry counts only if
        # it was taken on the kernels that are running now (its kernel_sha = the hash of today's device
        # sources); traffic_source says which entry, at which git head, or why there is none.
        traffic, traffic_source = None, {"file": "profiles/traffic.json", "key": "%s_%s" % (workload, args.pcm)}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        rec = json.load(open(tpath)).get(traffic_source["key"]) if os.path.exists(tpath) else None
        if rec is None:
            traffic_source["status"] = "no entry"
        elif rec.get("kernel_sha") != pkg.kernel_source_sha():
            traffic_source.update(status="stale: measured on other kernel sources", head=rec.get("head"),
                                  kernel_sha=rec.get("kernel_sha"), now=pkg.kernel_source_sha())
        else:
            traffic = rec["bytes_per_frame"] * n
            traffic_source.update(status="ok", head=rec.get("head"), kernel_sha=rec["kernel_sha"],
                                  bytes_per_frame=rec["bytes_per_frame"])
        out = {
            "metric": "HE-AAC frames/s (batched)", "value": value, "unit": "frames/s",
            "n_gpus": world, "per_gpu_value": value / world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg_desc, "frames_per_gpu": n, "pcm": args.pcm,
                       "bytes_per_frame": bytes_per_frame, "flops_per_frame": ALGO_FLOPS[workload],
                       "parallelism": "frames sharded by index, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel_ms": kern_ms,
                         # second ceiling: plain f32 add/mul, no FMA by construction
                         "gflops": ALGO_FLOPS[workload] * n / (kern_ms * 1e-3) / 1e9,
                         "gflops_peak_nofma": VALU_NOFMA_PEAK_GFLOPS},
        }
        if not dry and cfg != pkg.CFG_LC_STEREO:
            # how many bands of the X rows the HF / PS stage stored for the timed batch (the rest is +0 and neither
            # written nor fetched): the headline leans on this share, so it is part of the workload's description
            out["config"]["x_bands_stored"] = dev.x_bands_shares(n)
        if pkg.LIB_OVERRIDDEN:
            out["library"] = pkg.LIB_PATH          # a variant build (HEAAC_LIB_PATH), not the product library
        if dry:
            out["dry_run"] = True
        if gather_ms is not None:
            out["pcm_gather_ms"] = gather_ms
        if gather_check is not None:
            out["pcm_gather_check"] = gather_check
        if world == 1 and not args.no_cpu_baseline and not dry:
            out["cpu_baseline"] = cpu_baseline(pkg, synth, cfg)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
