"""Synthetic inputs for the HE-AAC DSP path: seeded, shaped like
what the reference's parsers hand to spectral_to_sample(), valid by construction.

Host logic only (numpy); no GPU, no oracle.  Generators are time-major: each
step yields the parameters of one frame for `n` independent streams, so state
chains from step to step exactly as in a serial decode.
"""
import numpy as np

from . import (ICS_DT, SBR_FRAME_DT, SBR_HDR_DT, PS_FRAME_DT, CFG_HEV1, CFG_HEV2, CFG_HEV1_MONO,
               ONLY_LONG_SEQUENCE, LONG_START_SEQUENCE, EIGHT_SHORT_SEQUENCE, LONG_STOP_SEQUENCE)

SF_SCALE = 1.0 / (1024.0 * 32768.0)     # |ac->sf_scale| on the C path, aacdec.c:575


class _IcsChain:
    """Window-sequence state machine: 90 % ONLY_LONG, else START -> EIGHT_SHORT -> STOP."""

    def __init__(self, rng, n, p_transition=0.1):
        self.rng, self.n, self.p = rng, n, p_transition
        self.ws_prev = np.zeros(n, np.uint8)
        self.kb_prev = np.zeros(n, np.uint8)
        self.phase = np.zeros(n, np.int8)      # 0 idle, 1 after START, 2 after SHORT

    def step(self):
        rng, n = self.rng, self.n
        ws = np.zeros(n, np.uint8)
        start = (self.phase == 0) & (rng.random(n) < self.p)
        ws[start] = LONG_START_SEQUENCE
        ws[self.phase == 1] = EIGHT_SHORT_SEQUENCE
        # stay short for a second frame now and then
        again = (self.phase == 2) & (rng.random(n) < 0.3)
        ws[again] = EIGHT_SHORT_SEQUENCE
        ws[(self.phase == 2) & ~again] = LONG_STOP_SEQUENCE
        new_phase = np.zeros(n, np.int8)
        new_phase[start] = 1
        new_phase[self.phase == 1] = 2
        new_phase[again] = 2
        kb = rng.integers(0, 2, n).astype(np.uint8)
        ics = np.zeros(n, ICS_DT)
        ics["window_sequence"][:, 0] = ws
        ics["window_sequence"][:, 1] = self.ws_prev
        ics["use_kb_window"][:, 0] = kb
        ics["use_kb_window"][:, 1] = self.kb_prev
        self.ws_prev, self.kb_prev, self.phase = ws, kb, new_phase
        return ics


def _coeffs(rng, ics, bins):
    """Uniform +-4096*|sf_scale| coefficients, band-limited to `bins` of 1024
    (per 128-bin window for EIGHT_SHORT frames)."""
    n = ics.shape[0]
    c = ((rng.random((n, 1024), dtype=np.float32) * 2 - 1) * np.float32(4096 * SF_SCALE)).astype(np.float32)
    if bins < 1024:
        mask_long = np.arange(1024) < bins
        mask_short = (np.arange(1024) % 128) < max(1, bins // 8)
        short = ics["window_sequence"][:, 0] == EIGHT_SHORT_SEQUENCE
        c *= np.where(short[:, None], mask_short[None, :], mask_long[None, :]).astype(np.float32)
    return c


def lc_stream(rng, n, steps, channels=2):
    """Yield (coeffs [n][channels][1024] f32, ics [n][channels]) per step."""
    chains = [_IcsChain(rng, n) for _ in range(channels)]
    for _ in range(steps):
        ics = np.stack([ch.step() for ch in chains], axis=1)
        coeffs = np.stack([_coeffs(rng, ics[:, c], 1024) for c in range(channels)], axis=1)
        yield np.ascontiguousarray(coeffs), np.ascontiguousarray(ics)


# ---------------------------------------------------------------------------
# HE-AAC (SBR + PS) parameter streams
# ---------------------------------------------------------------------------
def null_header(pkg):
    """The decoder's SBR state before any header has arrived: kx = 32, m = 0 (aacsbr.c:130), nothing
    else set.  Frames that refer to it carry start = 0 ("pure upsampling": analysis, the low band
    copied through, synthesis)."""
    h = np.zeros(1, pkg.SBR_HDR_DT)
    h["kx"] = 32
    h["map_hi"] = h["map_lo"] = h["map_nq"] = h["map_lim"] = h["map_mid"] = h["map_src"] = 0xff
    return h


def default_headers(pkg, extra=False, null=False):
    """SBR header table.  Entry 0 is the survey's probe header (48 kHz SBR rate:
    k0=13, kx=13, m=32, 3 patches); `extra` adds variants (smoothing on,
    non-interpolated envelopes, limiter off, other band layouts); `null` appends null_header()
    as the LAST entry."""
    hs = [pkg.sbr_make_header()]
    if extra:
        hs += [
            pkg.sbr_make_header(smoothing_mode=0),
            pkg.sbr_make_header(interpol_freq=0, limiter_bands=1, limiter_gains=0),
            pkg.sbr_make_header(start_freq=2, stop_freq=8, xover=2, freq_scale=0, alter_scale=1,
                                noise_bands=1, limiter_bands=2, limiter_gains=3, smoothing_mode=0, amp_res=0),
            pkg.sbr_make_header(start_freq=2, stop_freq=7, xover=0, freq_scale=0, alter_scale=0,
                                noise_bands=2, limiter_bands=0, limiter_gains=1),
            pkg.sbr_make_header(start_freq=0, stop_freq=3, xover=1, freq_scale=0, alter_scale=1,
                                noise_bands=3, limiter_bands=2, interpol_freq=0, smoothing_mode=0),
            pkg.sbr_make_header(start_freq=2, stop_freq=1, xover=0, freq_scale=3, alter_scale=0,
                                noise_bands=3, limiter_bands=3),
        ]
    if null:
        hs.append(null_header(pkg))
    return np.concatenate(hs)


class _SbrChain:
    """Per-stream SBR grid/envelope state machine producing what read_sbr_grid(),
    read_sbr_envelope(), read_sbr_noise(), read_sbr_invf() and
    read_sbr_sinusoidal_coding() leave in SBRData (aacsbr.c:609-889)."""

    def __init__(self, rng, hdr, hdr_idx, varfrac=0.15):
        self.rng, self.hdr, self.hdr_idx, self.varfrac = rng, hdr, hdr_idx, varfrac
        self.first = True
        self.num_env = 0
        self.t_env_last = 16        # t_env[bs_num_env] of the previous frame
        self.t_env = [0] * 8
        self.freq_res_last = 0
        self.e_a1 = -1
        self.invf = np.zeros(5, np.uint8)

    def step(self, out):
        rng, h = self.rng, self.hdr
        num_env_old = self.num_env
        t_old = self.t_env_last if not self.first else 0
        out["t_env_num_env_old"] = t_old
        out["bs_freq_res"][0] = self.freq_res_last
        carry = max(t_old - 16, 0)
        pointer = 0
        if carry == 0 and rng.random() >= self.varfrac:
            cls = 0                                              # FIXFIX
            num_env = int(rng.choice([1, 2, 4]))
            t = [0] + [((16 + (num_env >> 1)) // num_env) * (i + 1) for i in range(num_env - 1)] + [16]
            fr = [int(rng.integers(0, 2))] * num_env
            amp = 0 if num_env == 1 else int(h["bs_amp_res_header"])
        elif carry == 0:
            cls = 1                                              # FIXVAR
            trail = 16 + int(rng.integers(0, 4))
            nrt = int(rng.integers(0, 4))
            num_env = nrt + 1
            t = [0] * (num_env + 1)
            t[num_env] = trail
            for i in range(nrt):
                t[num_env - 1 - i] = t[num_env - i] - 2 * int(rng.integers(0, 2)) - 2
            pointer = int(rng.integers(0, num_env + 2))
            fr = [int(x) for x in rng.integers(0, 2, num_env)]
            amp = int(h["bs_amp_res_header"])
        else:
            cls = 2                                              # VARFIX
            nrl = int(rng.integers(0, 4))
            num_env = nrl + 1
            t = [carry] + [0] * num_env
            for i in range(nrl):
                t[i + 1] = t[i] + 2 * int(rng.integers(0, 2)) + 2
            t[num_env] = 16
            pointer = int(rng.integers(0, num_env + 2))
            fr = [int(x) for x in rng.integers(0, 2, num_env)]
            amp = int(h["bs_amp_res_header"])
        out["bs_num_env"] = num_env
        out["bs_amp_res"] = amp
        self.t_env[: num_env + 1] = t                           # entries behind the last border persist (uint8_t t_env[8])
        out["t_env"][:] = self.t_env
        out["bs_freq_res"][1: num_env + 1] = fr
        num_noise = 2 if num_env > 1 else 1
        out["bs_num_noise"] = num_noise
        tq = [t[0], t[num_env]]
        if num_noise > 1:
            if cls == 0:
                idx = num_env >> 1
            elif cls == 1:
                # unsigned bs_pointer (aacsbr.c:613, 729): pointer 0 selects the stale entry behind the last border
                idx = num_env + 1 if pointer == 0 else num_env - max(pointer - 1, 1)
            else:
                idx = 1 if pointer == 0 else (num_env - 1 if pointer == 1 else pointer - 1)
            tq = [t[0], self.t_env[idx], t[num_env]]
        out["t_q"][: len(tq)] = tq
        # l_APrev / l_A (aacsbr.c:737-743)
        e_a0 = -1 if (self.e_a1 != num_env_old) else 0
        e_a1 = -1
        if cls == 1 and pointer:
            e_a1 = num_env + 1 - pointer
        elif cls == 2 and pointer > 1:
            e_a1 = pointer - 1
        out["e_a"][:] = [e_a0, e_a1]
        # scalefactors, still integers (dequantised on the GPU)
        lo, hi = (8, 24) if amp else (16, 47)
        out["env_facs_q"][:num_env, :] = rng.integers(lo, hi, (num_env, 48))
        out["noise_facs_q"][:num_noise, :] = rng.integers(4, 12, (num_noise, 5))
        out["bs_invf_mode"][1] = self.invf
        out["bs_invf_mode"][0] = rng.integers(0, 4, 5)
        if rng.random() < 0.25:
            out["bs_add_harmonic_flag"] = 1
            out["bs_add_harmonic"][:] = rng.random(48) < 0.15
        self.first = False
        self.num_env, self.t_env_last, self.e_a1 = num_env, t[num_env], e_a1
        self.freq_res_last = fr[-1]
        self.invf = out["bs_invf_mode"][0].copy()


class _PsChain:
    def __init__(self, rng, mode):
        self.rng, self.mode = rng, mode        # mode: "20", "34", "mix"
        self.num_env = 0
        self.is34 = 0

    def step(self, out):
        rng = self.rng
        out["start"] = 1
        out["num_env_old"] = self.num_env
        out["is34bands_old"] = self.is34
        if self.mode == "20":
            layout = 1
        elif self.mode == "34":
            layout = 2
        else:
            layout = int(rng.integers(0, 3))
        nr = [10, 20, 34][layout]
        fine = int(rng.integers(0, 2)) if self.mode != "20" else 0
        out["nr_iid_par"] = nr
        out["nr_icc_par"] = nr if self.mode != "mix" else [10, 20, 34][int(rng.integers(0, 3))]
        out["nr_ipdopd_par"] = [5, 11, 17][layout]
        out["iid_quant"] = fine
        out["icc_mode"] = 1 if self.mode == "20" else int(rng.choice([0, 1, 2, 3, 4, 5]))
        out["enable_ipdopd"] = 0 if self.mode == "20" else 1
        is34 = int(out["nr_iid_par"] == 34 or out["nr_icc_par"] == 34)
        out["is34bands"] = is34
        if self.mode == "20" or rng.random() < 0.7:
            num_env = int(rng.choice([1, 2, 4]))
            border = [-1] + [(e * 32 >> {1: 0, 2: 1, 4: 2}[num_env]) - 1 for e in range(1, num_env + 1)]
        else:
            # frame_class 1: explicit borders; the parser's fix-up appends one ending at 31
            num_env = int(rng.integers(1, 5))
            cuts = sorted(rng.choice(np.arange(1, 31), size=num_env, replace=False).tolist())
            border = [-1] + cuts
            if border[-1] < 31:
                num_env += 1
                border.append(31)
        out["num_env"] = num_env
        out["border_position"][: num_env + 1] = border
        lim = 15 if fine else 7
        out["iid_par"][:num_env, :nr] = rng.integers(-lim, lim + 1, (num_env, nr))
        out["icc_par"][:num_env, : int(out["nr_icc_par"])] = rng.integers(0, 8, (num_env, int(out["nr_icc_par"])))
        if out["enable_ipdopd"]:
            out["ipd_par"][:num_env, : int(out["nr_ipdopd_par"])] = rng.integers(0, 8, (num_env, int(out["nr_ipdopd_par"])))
            out["opd_par"][:num_env, : int(out["nr_ipdopd_par"])] = rng.integers(0, 8, (num_env, int(out["nr_ipdopd_par"])))
        self.num_env, self.is34 = num_env, is34


def he_stream(rng, cfg, n, steps, hdr, ps_mode="20", hdr_choice=None, core_bins=400, coupling=0.0,
              events=None):
    """Yield dicts {coeffs, ics, sbr, ps} per step for n streams.
    hdr: header table; hdr_choice: per-stream header index (default all 0);
    coupling: fraction of CPE streams coded with bs_coupling = 1 (HE-AACv1 only).
    events: the decoder's degrade and transition paths, all off by default --
      lead_in   : up to this many frames per stream BEFORE the first SBR header (start = 0 on the null
                  header, which must be the table's last entry: ff_sbr_apply's pure-upsampling path,
                  aacsbr.c:1723-1750 with kx = 32, m = 0)
      p_switch  : per frame, the stream's header changes to another table entry (sbr->reset = 1,
                  kx[0] / m[0] = the old header's, aacsbr.c:1062-1073, 1412-1446, 1632-1637)
      p_drop    : per frame, the SBR payload is unusable (start = 0 mid-stream, aacsbr.c:989-1000): no HF
                  stage, state passes through; the grid of the frame is never parsed
      p_ps_off  : per frame, the PS payload is unusable (ps->start = 0, aacps.c:275-277): the mono QMF
                  signal is copied to both channels (aacsbr.c:1755), PS state untouched."""
    ncore = 2 if cfg == CFG_HEV1 else 1
    ev = dict(lead_in=0, p_switch=0.0, p_drop=0.0, p_ps_off=0.0)
    ev.update(events or {})
    nreal = len(hdr) - 1 if ev["lead_in"] else len(hdr)       # the null header is not a switch target
    if hdr_choice is None:
        hdr_choice = np.zeros(n, int)
    hdr_choice = np.array(hdr_choice, int).copy()
    ics_chains = [_IcsChain(rng, n) for _ in range(ncore)]
    sbr_chains = [[_SbrChain(rng, hdr[hdr_choice[s]], hdr_choice[s]) for _ in range(ncore)] for s in range(n)]
    ps_chains = [_PsChain(rng, ps_mode) for _ in range(n)] if cfg == CFG_HEV2 else None
    coupled = (rng.random(n) < coupling) if cfg == CFG_HEV1 else np.zeros(n, bool)
    lead = rng.integers(0, ev["lead_in"] + 1, n) if ev["lead_in"] else np.zeros(n, int)
    started = np.zeros(n, bool)                               # a header has been seen
    prev_kx_m = [(32, 0)] * n                                 # kx[1] / m[1] of the previous frame
    for step in range(steps):
        ics = np.stack([ch.step() for ch in ics_chains], axis=1)
        coeffs = np.stack([_coeffs(rng, ics[:, c], core_bins) for c in range(ncore)], axis=1)
        sbr = np.zeros(n, SBR_FRAME_DT)
        ps = np.zeros(n, PS_FRAME_DT) if cfg == CFG_HEV2 else None
        for s in range(n):
            fr = sbr[s]
            fr["kx_old"], fr["m_old"] = prev_kx_m[s]
            if step < lead[s]:
                # no header yet: kx = 32, m = 0, nothing parsed
                fr["hdr"] = len(hdr) - 1
                fr["start"] = 0
                prev_kx_m[s] = (32, 0)
            else:
                reset = not started[s]
                if started[s] and nreal > 1 and ev["p_switch"] > 0 and rng.random() < ev["p_switch"]:
                    hdr_choice[s] = (hdr_choice[s] + 1 + int(rng.integers(0, nreal - 1))) % nreal
                    for c in range(ncore):
                        sbr_chains[s][c].hdr, sbr_chains[s][c].hdr_idx = hdr[hdr_choice[s]], hdr_choice[s]
                    reset = True
                h = hdr[hdr_choice[s]]
                fr["hdr"] = hdr_choice[s]
                fr["reset"] = 1 if reset else 0
                prev_kx_m[s] = (int(h["kx"]), int(h["m"]))
                if started[s] and not reset and ev["p_drop"] > 0 and rng.random() < ev["p_drop"]:
                    fr["start"] = 0
                    for c in range(ncore):
                        ch = sbr_chains[s][c]
                        fr["ch"][c]["t_env_num_env_old"] = 0 if ch.first else ch.t_env_last
                else:
                    fr["start"] = 1
                    started[s] = True
                    for c in range(ncore):
                        sbr_chains[s][c].step(fr["ch"][c])
            if fr["start"] and coupled[s]:
                # read_sbr_channel_pair_element with bs_coupling (aacsbr.c:842-858): channel 1 takes
                # channel 0's grid (copy_sbr_grid, :747-766) and inverse-filtering modes; envelope and
                # noise data are read per channel (balance values for channel 1)
                fr["bs_coupling"] = 1
                c0, c1 = fr["ch"][0], fr["ch"][1]
                keep_prev = (int(c1["bs_freq_res"][0]), int(c1["t_env_num_env_old"]), int(c1["e_a"][0]))
                for f in ("bs_num_env", "bs_num_noise", "bs_amp_res", "bs_freq_res", "t_env", "t_q", "e_a"):
                    c1[f] = c0[f]
                c1["bs_freq_res"][0], c1["t_env_num_env_old"], c1["e_a"][0] = keep_prev
                c1["bs_invf_mode"][0] = c0["bs_invf_mode"][0]
                lo, hi = (0, 25) if c0["bs_amp_res"] else (0, 49)          # balance: pan_offset +- 12 / 24
                c1["env_facs_q"][: int(c0["bs_num_env"]), :] = rng.integers(lo, hi, (int(c0["bs_num_env"]), 48))
                c1["noise_facs_q"][: int(c0["bs_num_noise"]), :] = rng.integers(0, 25, (int(c0["bs_num_noise"]), 5))
                ch1 = sbr_chains[s][1]
                ch0 = sbr_chains[s][0]
                ch1.num_env, ch1.t_env_last, ch1.e_a1 = ch0.num_env, ch0.t_env_last, ch0.e_a1
                ch1.freq_res_last, ch1.invf = ch0.freq_res_last, c1["bs_invf_mode"][0].copy()
            if ps is not None:
                if ev["p_ps_off"] > 0 and rng.random() < ev["p_ps_off"]:
                    ps[s]["start"] = 0
                else:
                    ps_chains[s].step(ps[s])
        yield dict(coeffs=np.ascontiguousarray(coeffs), ics=np.ascontiguousarray(ics), sbr=sbr, ps=ps)


# ---------------------------------------------------------------------------
# Spectral tools (M/S, intensity stereo, TNS): side info as the parser leaves it.
# Band layouts are the 48 kHz ones (ISO/IEC 14496-3 Tables 4.129 / 4.130 =
# swb_offset_1024_48 / swb_offset_128_48, tns_max_bands 40 / 14; aactab.c:1107-1120, 1200-1206).
# ---------------------------------------------------------------------------
SWB_1024_48 = [0, 4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 48, 56, 64, 72, 80, 88, 96, 108, 120, 132, 144, 160,
               176, 196, 216, 240, 264, 292, 320, 352, 384, 416, 448, 480, 512, 544, 576, 608, 640, 672, 704,
               736, 768, 800, 832, 864, 896, 928, 1024]
SWB_128_48 = [0, 4, 8, 12, 16, 20, 28, 36, 44, 56, 68, 80, 96, 112, 128]


def _tools_ics(rng, ics, short):
    off = SWB_128_48 if short else SWB_1024_48
    ics["num_windows"] = 8 if short else 1
    ics["num_swb"] = len(off) - 1
    ics["tns_max_bands"] = 14 if short else 40
    ics["max_sfb"] = rng.integers(4, len(off)) if rng.random() < 0.8 else len(off) - 1
    ics["swb_offset"][: len(off)] = off
    if short:
        # a random grouping of the eight windows (scale_factor_grouping)
        cuts = np.flatnonzero(rng.random(7) < 0.4) + 1
        edges = np.concatenate(([0], cuts, [8]))
        lens = np.diff(edges)
        ics["num_window_groups"] = len(lens)
        ics["group_len"][: len(lens)] = lens
    else:
        ics["num_window_groups"] = 1
        ics["group_len"][0] = 1


def tools_frames(rng, pkg, n, channels=2):
    """n HeaacToolsFrame records with every tool exercised (and the degenerate cases: zero-length
    and zero-order filters, ranges clipped by max_sfb / tns_max_bands, both directions)."""
    t = np.zeros(n, pkg.TOOLS_FRAME_DT)
    for f in range(n):
        fr = t[f]
        short = rng.random() < 0.3
        fr["common_window"] = channels == 2 and rng.random() < 0.8
        fr["ms_present"] = rng.integers(0, 3) if channels == 2 else 0
        fr["ms_mask"] = 1 if fr["ms_present"] == 2 else (rng.random(128) < 0.5)
        _tools_ics(rng, fr["ch"][0]["ics"], short)
        if fr["common_window"]:
            fr["ch"][1]["ics"] = fr["ch"][0]["ics"]
        else:
            _tools_ics(rng, fr["ch"][1]["ics"], rng.random() < 0.3)
        for c in range(channels):
            ch = fr["ch"][c]
            nb = int(ch["ics"]["num_window_groups"]) * int(ch["ics"]["max_sfb"])
            r = rng.random(128)
            bt = rng.integers(1, 12, 128)                 # spectral codebooks
            bt[r < 0.10] = 0                              # ZERO_BT
            bt[(r >= 0.10) & (r < 0.15)] = 13             # NOISE_BT
            if c == 1:
                bt[(r >= 0.15) & (r < 0.25)] = 14         # INTENSITY_BT2
                bt[(r >= 0.25) & (r < 0.35)] = 15         # INTENSITY_BT
            ch["band_type"] = bt
            ch["band_type"][nb:] = 0
            ch["sf"] = np.exp2(rng.integers(-12, 8, 128) / 4.0).astype(np.float32)
            pr = ch["pred"]
            pr["pred_sfb_max"] = 40                       # ff_aac_pred_sfb_max[3] (48 kHz), aactab.c:47-49
            pr["predictor_present"] = rng.random() < 0.7
            pr["prediction_used"][:41] = rng.random(41) < 0.6
            pr["predictor_reset_group"] = rng.integers(1, 31) if rng.random() < 0.15 else 0
            tns = ch["tns"]
            tns["present"] = rng.random() < 0.5
            nw = int(ch["ics"]["num_windows"])
            for w in range(nw):
                tns["n_filt"][w] = rng.integers(0, 2 if nw == 8 else 4)
                for k in range(int(tns["n_filt"][w])):
                    tns["length"][w][k] = rng.integers(0, int(ch["ics"]["num_swb"]) + 1)
                    tns["order"][w][k] = rng.integers(0, 8 if nw == 8 else 13) if rng.random() < 0.9 else 20
                    tns["direction"][w][k] = rng.integers(0, 2)
                    q = rng.integers(-8, 8, 20)
                    tns["coef"][w][k] = np.sin(q * np.pi / 17.0).astype(np.float32)
    return t
