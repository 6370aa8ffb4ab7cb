import sys, os, importlib
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
pkg = importlib.import_module("ffmpeg-heaac_amd")
import oracle_lib as oracle
import sbr_bitwriter as SW
import test_sbr_parse as T
cpe = False
rng = np.random.default_rng(71 + cpe)
n = 32; ch = 1; hcfg = pkg.CFG_HEV2
cfg = T._he_cfg(pkg, ch, True)
tab = pkg.SbrHeaderTable(64)
st = np.zeros(n, pkg.AAC_STREAM_DT); sst = pkg.sbr_streams(n)
writers = [SW.SbrStreamWriter(pkg, ch, ps=True) for _ in range(n)]
state = np.zeros((n, pkg.STATE_WORDS[hcfg]), np.float32)
d_state = torch.from_numpy(state).cuda()
dev = pkg.Device()
d_rng = torch.full((n,), 0x1f2e3d4c, dtype=torch.int32, device="cuda"); ref_rng = np.full(n, 0x1f2e3d4c, np.int32)
for step in range(5):
    aus, exps = T._he_units(pkg, rng, writers, cpe, new_header=step == 2)
    out = pkg.heaac_parse_batch(cfg, st, sst, tab, aus, with_ps=True)
    hdr = tab.headers()
    exp_sbr = np.concatenate([e["sbr"] for e in exps]); exp_sbr["hdr"] = out["sbr"]["hdr"]
    exp_ps = np.concatenate([e["ps"] for e in exps])
    coeffs = np.ascontiguousarray(out["coeffs"][:, :ch])
    ref_c, ref_rng = oracle.spectral_tools_batch(ch, coeffs, out["tools"], rng=ref_rng)
    ics = np.ascontiguousarray(out["ics"][:, :ch])
    scale = (2.0 ** -np.ceil(np.log2(np.maximum(np.abs(ref_c).max(axis=(1, 2)), 1.0)))).astype(np.float32)
    ref_c = ref_c * scale[:, None, None]
    state_in = state
    ref_pcm, state = oracle.he_decode_batch(hcfg, ref_c, ics, exp_sbr, hdr, exp_ps, state, pkg.PCM_F32)
    d_c = torch.from_numpy(coeffs).cuda()
    dev.spectral_tools(ch, d_c, pkg.to_device(out["tools"]), rng=d_rng)
    d_c.mul_(torch.from_numpy(scale).cuda()[:, None, None])
    print("scale", scale.min(), scale.max(), "finite", bool(np.isfinite(ref_c).all()))
    pcm, d_state = dev.he_decode(hcfg, d_c, pkg.to_device(ics), pkg.to_device(out["sbr"]), pkg.to_device(hdr), pkg.to_device(out["ps"]), d_state)
    got = pcm.cpu().numpy()
    bad = (got.view(np.uint32) != ref_pcm.view(np.uint32))
    sbad = (d_state.cpu().numpy().view(np.uint32) != state.view(np.uint32))
    print("step", step, "pcm bad", int(bad.sum()), "streams", np.nonzero(bad.any(axis=(1, 2)))[0].tolist())
    off = [0, 512, 512 + 1972, 512 + 1972 + 1152, 512 + 1972 + 2304, pkg.STATE_WORDS[hcfg]]
    for s in np.nonzero(bad.any(axis=(1, 2)) | sbad.any(axis=1))[0]:
        p = out["ps"][s]; f = out["sbr"][s]
        print("  stream", s, "chan bad", bad[s].sum(axis=1).tolist(), "first", np.nonzero(bad[s].any(axis=0))[0][:3].tolist(),
              "state bad", [int(sbad[s, off[i]:off[i + 1]].sum()) for i in range(5)],
              "maxdiff", float(np.nanmax(np.abs(got[s] - ref_pcm[s]))),
              "ps", dict(start=int(p["start"]), E=int(p["num_env"]), Eold=int(p["num_env_old"]), is34=int(p["is34bands"]), old34=int(p["is34bands_old"]),
                         ipd=int(p["enable_ipdopd"]), q=int(p["iid_quant"]), icc=int(p["icc_mode"]), ni=int(p["nr_iid_par"]), nc=int(p["nr_icc_par"]), np_=int(p["nr_ipdopd_par"]),
                         border=p["border_position"][:6].tolist()),
              "sbr", dict(start=int(f["start"]), reset=int(f["reset"]), L=int(f["ch"][0]["bs_num_env"]), t=f["ch"][0]["t_env"][:6].tolist(), hdr=int(f["hdr"])))
    for s in np.nonzero(bad.any(axis=(1, 2)) | sbad.any(axis=1))[0][:2]:
        ds = d_state.cpu().numpy()
        w = np.nonzero(sbad[s, off[4]:])[0]
        print("  ps state words", w.tolist(), "gpu", ds[s, off[4]:][w].tolist(), "ref", state[s, off[4]:][w].tolist())
        for c in range(2):
            w = np.nonzero(bad[s, c])[0]
            print("  pcm ch", c, "bad range", int(w.min()), int(w.max()), "count", len(w))
        np.savez("gpurun_out/bad_frame_%d_%d.npz" % (step, s), coeffs=ref_c[s:s + 1], ics=ics[s:s + 1], sbr=out["sbr"][s:s + 1],
                 ps=out["ps"][s:s + 1], hdr=hdr, state_in=state_in[s:s + 1], gpu_pcm=got[s:s + 1], gpu_state=ds[s:s + 1])
    # keep going with the oracle's state on both sides so later steps are judged on their own
    d_state = torch.from_numpy(state).cuda()
