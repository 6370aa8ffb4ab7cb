"""Randomised GPU-vs-oracle parity soak of the spectral tools (noise substitution, AAC-Main prediction, M/S, intensity,
TNS; ffmpeg-heaac_amd/synth.py tools_frames: every tool, degenerate filters, both directions, orders up to 20), generator
and predictor state chained over the steps of a run.  usage: python tools/soak_tools.py [seconds]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
pkg = importlib.import_module("ffmpeg-heaac_amd")
synth = importlib.import_module("ffmpeg-heaac_amd.synth")
import oracle_lib as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
dev = pkg.Device(4096)
t0 = time.time()
runs = frames = nans = 0
seed = 0
while time.time() - t0 < budget:
    seed += 1
    rng = np.random.default_rng(seed)
    ch = int(rng.integers(1, 3))
    main = bool(rng.integers(0, 2))
    n = int(rng.choice([1, 37, 500, 1500]))
    rs = rng.integers(-2**31, 2**31, n).astype(np.int32)
    pred = np.tile(np.array([0, 0, 1, 1, 0, 0], np.float32), (n, ch * pkg.MAX_PREDICTORS, 1)) if main else None
    d_rs = torch.from_numpy(rs.copy()).cuda()
    d_pred = torch.from_numpy(pred.copy()).cuda() if main else None
    for step in range(3):
        tools = synth.tools_frames(rng, pkg, n, ch)
        coeffs = (rng.standard_normal((n, ch, 1024)) * 10.0 ** rng.uniform(-6, 2)).astype(np.float32)
        if main:
            ref, rs, pred = O.spectral_tools_batch(ch, coeffs, tools, rng=rs, pred=pred)
        else:
            ref, rs = O.spectral_tools_batch(ch, coeffs, tools, rng=rs)
        d = torch.from_numpy(coeffs).cuda()
        dev.spectral_tools(ch, d, pkg.to_device(tools), rng=d_rs, pred=d_pred)
        got = d.cpu().numpy()
        # bit for bit, except that a NaN is a NaN (an order-20 filter on loud noise runs away to infinity and inf - inf:
        # the host's invalid operations give the negative quiet NaN, the GPU's the positive one)
        def same(x, y):
            x, y = np.ascontiguousarray(x, np.float32).reshape(-1), np.ascontiguousarray(y, np.float32).reshape(-1)
            nx, ny = np.isnan(x), np.isnan(y)
            return np.array_equal(nx, ny) and np.array_equal(x.view(np.uint32)[~nx], y.view(np.uint32)[~ny])
        ok = same(got, ref) and np.array_equal(d_rs.cpu().numpy(), rs)
        nans += int(np.isnan(ref).any(axis=(1, 2)).sum())
        if main:
            ok = ok and same(d_pred.cpu().numpy(), pred)
        if not ok:
            bad = np.argwhere(got.view(np.uint32) != ref.view(np.uint32))
            print("MISMATCH seed %d step %d ch %d main %d n %d first %s" % (seed, step, ch, main, n, bad[:3].tolist()))
            sys.exit(1)
        frames += n
    runs += 1
print("soak ok: %d runs, %d frames (%d of them with a NaN somewhere), %.0f s" % (runs, frames, nans, time.time() - t0))
