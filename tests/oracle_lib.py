"""ctypes binding of oracle/_build/libheaac_oracle.so (test infrastructure).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Record layouts mirror include/heaac_dsp.h.
"""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "libheaac_oracle.so")

# ---- record dtypes (include/heaac_dsp.h) ----
ICS_DT = np.dtype([("window_sequence", "u1", (2,)), ("use_kb_window", "u1", (2,))])

SBR_HDR_DT = np.dtype([
    ("k0", "u1"), ("k2", "u1"), ("kx", "u1"), ("m", "u1"),
    ("n", "u1", (2,)), ("n_q", "u1"), ("n_lim", "u1"),
    ("n_master", "u1"), ("num_patches", "u1"), ("bs_limiter_gains", "u1"),
    ("bs_interpol_freq", "u1"), ("bs_smoothing_mode", "u1"), ("bs_amp_res_header", "u1"),
    ("pad0", "u1", (2,)),
    ("patch_num_subbands", "u1", (6,)), ("patch_start_subband", "u1", (6,)),
    ("f_tablenoise", "u1", (6,)), ("pad1", "u1", (2,)),
    ("f_tablelow", "u1", (28,)), ("f_tablehigh", "u1", (52,)), ("f_tablelim", "u1", (32,)),
    ("map_hi", "u1", (64,)), ("map_lo", "u1", (64,)), ("map_nq", "u1", (64,)),
    ("map_lim", "u1", (64,)), ("map_mid", "u1", (64,)), ("map_src", "u1", (64,)),
])
assert SBR_HDR_DT.itemsize == 532

SBR_CH_DT = np.dtype([
    ("bs_num_env", "u1"), ("bs_num_noise", "u1"), ("bs_amp_res", "u1"), ("bs_add_harmonic_flag", "u1"),
    ("bs_freq_res", "u1", (8,)), ("t_env", "u1", (8,)), ("t_q", "u1", (3,)),
    ("t_env_num_env_old", "u1"), ("e_a", "i1", (2,)),
    ("bs_invf_mode", "u1", (2, 5)), ("bs_add_harmonic", "u1", (48,)),
    ("env_facs_q", "u1", (5, 48)), ("noise_facs_q", "u1", (2, 5)), ("pad", "u1", (2,)),
])
assert SBR_CH_DT.itemsize == 336

SBR_FRAME_DT = np.dtype([
    ("hdr", "<u2"), ("start", "u1"), ("reset", "u1"), ("kx_old", "u1"), ("m_old", "u1"),
    ("bs_coupling", "u1"), ("pad", "u1"), ("ch", SBR_CH_DT, (2,)),
])
assert SBR_FRAME_DT.itemsize == 680

PS_FRAME_DT = np.dtype([
    ("start", "u1"), ("is34bands", "u1"), ("is34bands_old", "u1"), ("num_env", "u1"),
    ("num_env_old", "u1"), ("enable_ipdopd", "u1"), ("iid_quant", "u1"), ("icc_mode", "u1"),
    ("nr_iid_par", "u1"), ("nr_icc_par", "u1"), ("nr_ipdopd_par", "u1"), ("pad", "u1"),
    ("border_position", "i1", (8,)),
    ("iid_par", "i1", (5, 34)), ("icc_par", "i1", (5, 34)),
    ("ipd_par", "i1", (5, 17)), ("opd_par", "i1", (5, 17)), ("pad2", "u1", (2,)),
])
assert PS_FRAME_DT.itemsize == 532

# state record layout (32-bit words)
ST_SAVED, ST_SBR, ST_SYNTH, ST_PS = 512, 1972, 1152, 4500
SBR_XHIST, SBR_WTAIL, SBR_YTAIL, SBR_GTAIL, SBR_QTAIL = 0, 288, 800, 1568, 1760
SBR_BW, SBR_IDXNOISE, SBR_IDXSINE, SBR_SIDX = 1952, 1957, 1958, 1959
PS_INBUF, PS_DELAY, PS_APDELAY, PS_PEAK, PS_PSMOOTH, PS_PDIFF, PS_H, PS_HIST = \
    0, 60, 2608, 4108, 4142, 4176, 4210, 4482

CFG_LC_MONO, CFG_LC_STEREO, CFG_HEV1, CFG_HEV2, CFG_HEV1_MONO = 0, 1, 2, 3, 4
PCM_F32, PCM_S16, PCM_S16_SSE2 = 0, 1, 2

STATE_WORDS = {
    CFG_LC_MONO: ST_SAVED,
    CFG_LC_STEREO: 2 * ST_SAVED,
    CFG_HEV1: 2 * ST_SAVED + 2 * ST_SBR + 2 * ST_SYNTH,
    CFG_HEV1_MONO: ST_SAVED + ST_SBR + ST_SYNTH,
    CFG_HEV2: ST_SAVED + ST_SBR + 2 * ST_SYNTH + ST_PS,
}
CORE_CH = {CFG_LC_MONO: 1, CFG_LC_STEREO: 2, CFG_HEV1: 2, CFG_HEV2: 1, CFG_HEV1_MONO: 1}
OUT_CH = {CFG_LC_MONO: 1, CFG_LC_STEREO: 2, CFG_HEV1: 2, CFG_HEV2: 2, CFG_HEV1_MONO: 1}
OUT_LEN = {CFG_LC_MONO: 1024, CFG_LC_STEREO: 1024, CFG_HEV1: 2048, CFG_HEV2: 2048, CFG_HEV1_MONO: 2048}

_lib = None


def build(force=False):
    """Compile the oracle (gcc, seconds)."""
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
    inc = os.path.join(ROOT, "include")
    srcs += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    if force or not os.path.exists(LIB_PATH) or any(
            os.path.getmtime(f) > os.path.getmtime(LIB_PATH) for f in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.oracle_tables.restype = C.c_void_p
        _lib.oracle_get_table.argtypes = [C.c_char_p, C.c_void_p, C.c_int]
        _lib.oracle_float_to_int16_one.argtypes = [C.c_float]
        _lib.oracle_tables()
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def get_table(name, n=4096):
    buf = np.zeros(n, np.float32)
    r = lib().oracle_get_table(name.encode(), _p(buf), n)
    if r < 0:
        raise KeyError(name)
    return buf[:r].copy()


def fft_calc(nbits, z):
    z = np.ascontiguousarray(z, dtype=np.complex64).copy()
    lib().oracle_fft_calc(C.c_int(nbits), _p(z))
    return z


def imdct_half(which, x):
    x = _f32(x)
    out = np.zeros_like(x)
    flat_in = x.reshape(-1, x.shape[-1])
    flat_out = out.reshape(-1, x.shape[-1])
    for i in range(flat_in.shape[0]):
        lib().oracle_imdct_half(C.c_int(which), _p(flat_out[i]), _p(flat_in[i]))
    return out


def imdct_calc(which, x):
    x = _f32(x)
    out = np.zeros(2 * x.shape[-1], np.float32)
    lib().oracle_imdct_calc(C.c_int(which), _p(out), _p(x))
    return out


def qmf_analysis(inp, xhist, scale=32768.0):
    inp = _f32(inp)
    xhist = _f32(xhist).copy()
    W = np.zeros((32, 32, 2), np.float32)
    lib().oracle_qmf_analysis(_p(inp), _p(xhist), _p(W), C.c_float(scale))
    return W, xhist


def qmf_synthesis(X, v, scale=2.0 ** -15, bias=385.0):
    X = _f32(X)
    v = _f32(v).copy()
    out = np.zeros(2048, np.float32)
    lib().oracle_qmf_synthesis(_p(X), _p(v), _p(out), C.c_float(scale), C.c_float(bias))
    return out, v


def qmf_synthesis_ds(X, v, scale=2.0 ** -15, bias=385.0):
    X = _f32(X)
    v = _f32(v).copy()
    out = np.zeros(1024, np.float32)
    lib().oracle_qmf_synthesis_ds(_p(X), _p(v), _p(out), C.c_float(scale), C.c_float(bias))
    return out, v


def float_to_int16(a):
    a = _f32(a)
    f = lib().oracle_float_to_int16_one
    return np.array([f(C.c_float(float(v))) for v in a.ravel()], np.int16).reshape(a.shape)


def float_to_int16_interleave(planes, sse2=False):
    """planes: [channels][len] float32 -> [len][channels] int16 (ff_float_to_int16_interleave_c)."""
    planes = [_f32(p) for p in planes]
    ch, n = len(planes), planes[0].shape[0]
    ptrs = (C.c_void_p * ch)(*[p.ctypes.data for p in planes])
    out = np.zeros((n, ch), np.int16)
    lib().oracle_float_to_int16_interleave(_p(out), ptrs, C.c_long(n), C.c_int(ch), C.c_int(1 if sse2 else 0))
    return out


def lc_decode_batch(channels, coeffs, ics, state_in, pcm_format=PCM_F32):
    coeffs = _f32(coeffs)
    n = coeffs.shape[0]
    ics = np.ascontiguousarray(ics, dtype=ICS_DT)
    state_in = _f32(state_in)
    state_out = np.zeros_like(state_in)
    if pcm_format == PCM_F32:
        pcm = np.zeros((n, channels, 1024), np.float32)
    else:
        pcm = np.zeros((n, 1024, channels), np.int16)
    r = lib().oracle_lc_decode_batch(C.c_int(channels), _p(coeffs), _p(ics), _p(state_in),
                                     _p(state_out), _p(pcm), C.c_int(pcm_format), C.c_size_t(n))
    if r:
        raise RuntimeError("oracle_lc_decode_batch -> %d" % r)
    return pcm, state_out


COUPLING_DT = np.dtype([("gain", "<f4", (2,)), ("on", "u1", (2,)), ("pad", "u1", (2,))])


def couple_after_imdct_batch(channels, pcm, cce, coupling, s16=False):
    """apply_independent_coupling on a copy of pcm [n][channels][1024]; returns (pcm, int16 [n][1024][channels] or None)."""
    out = _f32(pcm).copy()
    n = out.shape[0]
    cce = _f32(cce)
    coupling = np.ascontiguousarray(coupling, dtype=COUPLING_DT)
    o16 = np.zeros((n, 1024, channels), np.int16) if s16 else None
    r = lib().oracle_couple_after_imdct_batch(C.c_int(channels), _p(out), _p(cce), _p(coupling),
                                              _p(o16) if s16 else None, C.c_size_t(n))
    if r:
        raise RuntimeError("oracle_couple_after_imdct_batch -> %d" % r)
    return out, o16


def spectral_tools_batch(channels, coeffs, tools, rng=None, pred=None):
    """(PNS if rng, AAC-Main prediction if pred,) M/S + intensity + TNS on a copy of coeffs.
    Returns coeffs, followed by rng_out and / or pred_out when those were given."""
    out = np.ascontiguousarray(coeffs, np.float32).copy()
    tools = np.ascontiguousarray(tools)
    rin = rout = pin = pout = None
    if rng is not None:
        rin = np.ascontiguousarray(rng, np.int32); rout = np.empty_like(rin)
    if pred is not None:
        pin = np.ascontiguousarray(pred, np.float32); pout = np.empty_like(pin)
    lib().oracle_spectral_tools_batch(C.c_int(channels), _p(out), _p(tools), _p(rin), _p(rout), _p(pin), _p(pout),
                                      C.c_size_t(out.shape[0]))
    res = (out,) + ((rout,) if rng is not None else ()) + ((pout,) if pred is not None else ())
    return res[0] if len(res) == 1 else res


TOOLS_PRE, TOOLS_POST, TOOLS_ALL = 1, 2, 3


def spectral_tools_batch_ex(channels, stages, coeffs, tools, rng=None, pred=None, cce=None, cce_coeffs=None):
    """The staged form (PRE / POST halves, dependent coupling in POST): returns (coeffs, rng_out, pred_out), the
    last two None when not given."""
    out = np.ascontiguousarray(coeffs, np.float32).copy()
    tools = np.ascontiguousarray(tools)
    rin = rout = pin = pout = None
    if rng is not None:
        rin = np.ascontiguousarray(rng, np.int32); rout = rin.copy()
    if pred is not None:
        pin = np.ascontiguousarray(pred, np.float32); pout = pin.copy()
    n_cce = 0
    if cce is not None:
        cce = np.ascontiguousarray(cce); cce_coeffs = np.ascontiguousarray(cce_coeffs, np.float32)
        n_cce = cce.shape[1]
        assert cce_coeffs.shape == (out.shape[0], n_cce, 1024)
    lib().oracle_spectral_tools_batch_ex(C.c_int(channels), C.c_int(stages), _p(out), _p(tools), _p(rin), _p(rout),
                                         _p(pin), _p(pout), _p(cce) if n_cce else None,
                                         _p(cce_coeffs) if n_cce else None, C.c_int(n_cce), C.c_size_t(out.shape[0]))
    return out, rout, pout


def he_decode_batch(cfg, coeffs, ics, sbr, hdr, ps, state_in, pcm_format=PCM_F32, downsampled=False):
    coeffs = _f32(coeffs)
    n = coeffs.shape[0]
    ics = np.ascontiguousarray(ics, dtype=ICS_DT)
    sbr = np.ascontiguousarray(sbr, dtype=SBR_FRAME_DT)
    hdr = np.ascontiguousarray(hdr, dtype=SBR_HDR_DT)
    if ps is not None:
        ps = np.ascontiguousarray(ps, dtype=PS_FRAME_DT)
    state_in = _f32(state_in)
    assert state_in.shape == (n, STATE_WORDS[cfg]), (state_in.shape, STATE_WORDS[cfg])
    state_out = np.zeros_like(state_in)
    length = 1024 if downsampled else 2048
    if pcm_format == PCM_F32:
        pcm = np.zeros((n, OUT_CH[cfg], length), np.float32)
    else:
        pcm = np.zeros((n, length, OUT_CH[cfg]), np.int16)
    r = lib().oracle_he_decode_batch_ex(C.c_int(cfg), C.c_int(1 if downsampled else 0), _p(coeffs), _p(ics), _p(sbr), _p(hdr),
                                     C.c_size_t(hdr.shape[0]), _p(ps), _p(state_in), _p(state_out),
                                     _p(pcm), C.c_int(pcm_format), C.c_size_t(n))
    if r:
        raise RuntimeError("oracle_he_decode_batch -> %d" % r)
    return pcm, state_out


def he_decode_debug(cfg, coeffs, ics, sbr, hdr, ps, state_in):
    """One frame with stage dumps; returns dict."""
    coeffs = _f32(coeffs)
    ics = np.ascontiguousarray(ics, dtype=ICS_DT)
    sbr = np.ascontiguousarray(sbr, dtype=SBR_FRAME_DT)
    hdr = np.ascontiguousarray(hdr, dtype=SBR_HDR_DT)
    if ps is not None:
        ps = np.ascontiguousarray(ps, dtype=PS_FRAME_DT)
    state_in = _f32(state_in)
    state_out = np.zeros_like(state_in)
    d = dict(
        pcm=np.zeros((OUT_CH[cfg], 2048), np.float32),
        W=np.zeros((2, 32, 32, 2), np.float32),
        Xlow=np.zeros((32, 40, 2), np.float32),
        Xhigh=np.zeros((64, 40, 2), np.float32),
        Y=np.zeros((38, 64, 2), np.float32),
        Xsbr=np.zeros((2, 2, 38, 64), np.float32),
        X=np.zeros((2, 2, 38, 64), np.float32),
    )
    r = lib().oracle_he_decode_debug(C.c_int(cfg), _p(coeffs), _p(ics), _p(sbr), _p(hdr), _p(ps),
                                     _p(state_in), _p(state_out), _p(d["pcm"]), _p(d["W"]),
                                     _p(d["Xlow"]), _p(d["Xhigh"]), _p(d["Y"]), _p(d["Xsbr"]), _p(d["X"]))
    if r:
        raise RuntimeError("oracle_he_decode_debug -> %d" % r)
    d["state_out"] = state_out
    return d


def sbr_make_header(sample_rate=48000, start_freq=5, stop_freq=9, xover=0, freq_scale=2,
                    alter_scale=1, noise_bands=2, limiter_bands=2, limiter_gains=2,
                    interpol_freq=1, smoothing_mode=1, amp_res=1):
    h = np.zeros(1, SBR_HDR_DT)
    r = lib().oracle_sbr_make_header(_p(h), sample_rate, start_freq, stop_freq, xover, freq_scale,
                                     alter_scale, noise_bands, limiter_bands, limiter_gains,
                                     interpol_freq, smoothing_mode, amp_res)
    if r:
        raise ValueError("invalid SBR header (%d)" % r)
    return h
