"""ffmpeg-heaac_amd -- MI355X-native HE-AAC decode DSP (host-side Python mirror).

The product is the C-ABI shared library `libheaac_amd.so` (include/heaac_dsp.h,
include/heaac_fft.h, include/heaac_codec.h): hand-written HIP kernels for
gfx950 plus C host code.  This module only binds it with ctypes so that tests
and bench.py can drive it; PyTorch supplies device memory and streams, nothing
else.  There is NO CPU fallback: if the library is missing, or no gfx950 device
is visible when a compute entry point is called, an exception is raised.

The directory name contains a hyphen; load it with
    import importlib.util  (see __graft_entry__.load_package)
or simply `from __graft_entry__ import load_package; heaac = load_package()`.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# HEAAC_LIB_PATH: measurement tooling (tools/abv.sh) points the binding at a variant build under ab/
# without touching the product library; unset everywhere else.  Never silently: the override is announced on
# stderr when the library is loaded, LIB_OVERRIDDEN says so to whoever reports numbers (bench.py prints it).
PRODUCT_LIB_PATH = os.path.join(_HERE, "libheaac_amd.so")
LIB_PATH = os.environ.get("HEAAC_LIB_PATH") or PRODUCT_LIB_PATH
LIB_OVERRIDDEN = os.path.realpath(LIB_PATH) != os.path.realpath(PRODUCT_LIB_PATH)

# ---- constants (include/heaac_dsp.h) ----
ONLY_LONG_SEQUENCE, LONG_START_SEQUENCE, EIGHT_SHORT_SEQUENCE, LONG_STOP_SEQUENCE = 0, 1, 2, 3
CFG_LC_MONO, CFG_LC_STEREO, CFG_HEV1, CFG_HEV2, CFG_HEV1_MONO = 0, 1, 2, 3, 4
PCM_F32, PCM_S16, PCM_S16_SSE2 = 0, 1, 2
ADD_BIAS = 385.0

ST_SAVED, ST_SBR, ST_SYNTH, ST_PS = 512, 1972, 1152, 4500
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

# Algorithmic HBM bytes per frame, SURVEY.md s8(d) / BASELINE.md s3 (f32 PCM out).
ALGO_BYTES = {CFG_LC_STEREO: 24584, CFG_HEV1: 83976, CFG_HEV2: 85284}

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
SBR_CH_DT = np.dtype([
    ("bs_num_env", "u1"), ("bs_num_noise", "u1"), ("bs_amp_res", "u1"), ("bs_add_harmonic_flag", "u1"),
    ("bs_freq_res", "u1", (8,)), ("t_env", "u1", (8,)), ("t_q", "u1", (3,)),
    ("t_env_num_env_old", "u1"), ("e_a", "i1", (2,)),
    ("bs_invf_mode", "u1", (2, 5)), ("bs_add_harmonic", "u1", (48,)),
    ("env_facs_q", "u1", (5, 48)), ("noise_facs_q", "u1", (2, 5)), ("pad", "u1", (2,)),
])
SBR_FRAME_DT = np.dtype([
    ("hdr", "<u2"), ("start", "u1"), ("reset", "u1"), ("kx_old", "u1"), ("m_old", "u1"),
    ("bs_coupling", "u1"), ("pad", "u1"), ("ch", SBR_CH_DT, (2,)),
])
PS_FRAME_DT = np.dtype([
    ("start", "u1"), ("is34bands", "u1"), ("is34bands_old", "u1"), ("num_env", "u1"),
    ("num_env_old", "u1"), ("enable_ipdopd", "u1"), ("iid_quant", "u1"), ("icc_mode", "u1"),
    ("nr_iid_par", "u1"), ("nr_icc_par", "u1"), ("nr_ipdopd_par", "u1"), ("pad", "u1"),
    ("border_position", "i1", (8,)),
    ("iid_par", "i1", (5, 34)), ("icc_par", "i1", (5, 34)),
    ("ipd_par", "i1", (5, 17)), ("opd_par", "i1", (5, 17)), ("pad2", "u1", (2,)),
])
TOOLS_ICS_DT = np.dtype([
    ("num_windows", "u1"), ("num_window_groups", "u1"), ("max_sfb", "u1"), ("num_swb", "u1"),
    ("tns_max_bands", "u1"), ("pad", "u1", (3,)), ("group_len", "u1", (8,)), ("swb_offset", "<u2", (64,)),
])
TNS_DT = np.dtype([
    ("present", "u1"), ("n_filt", "u1", (8,)), ("length", "u1", (8, 4)), ("direction", "u1", (8, 4)),
    ("order", "u1", (8, 4)), ("pad", "u1", (3,)), ("coef", "<f4", (8, 4, 20)),
])
PRED_DT = np.dtype([("predictor_present", "u1"), ("predictor_reset_group", "u1"), ("pred_sfb_max", "u1"),
                    ("pad", "u1"), ("prediction_used", "u1", (44,))])
PRED_STATE_DT = np.dtype([("cor0", "<f4"), ("cor1", "<f4"), ("var0", "<f4"), ("var1", "<f4"), ("r0", "<f4"), ("r1", "<f4")])
MAX_PREDICTORS = 672
TOOLS_CH_DT = np.dtype([("ics", TOOLS_ICS_DT), ("band_type", "u1", (128,)), ("sf", "<f4", (128,)), ("tns", TNS_DT),
                        ("pred", PRED_DT)])
TOOLS_FRAME_DT = np.dtype([
    ("common_window", "u1"), ("ms_present", "u1"), ("pad", "u1", (2,)), ("ms_mask", "u1", (128,)),
    ("ch", TOOLS_CH_DT, (2,)),
])
assert TOOLS_ICS_DT.itemsize == 144 and TNS_DT.itemsize == 2668 and TOOLS_FRAME_DT.itemsize == 7132
MAX_CCE, MAX_CCE_LINKS = 16, 4
CC_BEFORE_TNS, CC_BETWEEN_TNS_AND_IMDCT, CC_AFTER_IMDCT = 0, 1, 3
TOOLS_PRE, TOOLS_POST, TOOLS_ALL = 1, 2, 3
CCE_LINK_DT = np.dtype([("target_ch", "u1"), ("pad", "u1", (3,)), ("gain", "<f4", (120,))])
CCE_FRAME_DT = np.dtype([("present", "u1"), ("elem_id", "u1"), ("coupling_point", "u1"), ("n_links", "u1"),
                         ("behind_target", "u1"), ("seq", "u1"), ("outputs_before", "u1"), ("pad", "u1"), ("ics", TOOLS_ICS_DT), ("band_type", "u1", (128,)),
                         ("link", CCE_LINK_DT, (MAX_CCE_LINKS,))])
assert CCE_LINK_DT.itemsize == 484 and CCE_FRAME_DT.itemsize == 2216
assert SBR_HDR_DT.itemsize == 532 and SBR_CH_DT.itemsize == 336
assert SBR_FRAME_DT.itemsize == 680 and PS_FRAME_DT.itemsize == 532

# Every symbol include/*.h declares (checked by tests/test_abi.py).
EXPORTED = [
    # heaac_dsp.h
    "heaac_device_create", "heaac_device_destroy", "heaac_device_workspace_bytes",
    "heaac_strerror", "heaac_imdct_half_batch", "heaac_lc_decode_batch",
    "heaac_he_decode_batch", "heaac_he_decode_batch_ex", "heaac_qmf_analysis_batch", "heaac_qmf_synthesis_batch",
    "heaac_qmf_synthesis_ds_batch",
    "heaac_sbr_make_header", "heaac_build_info", "heaac_spectral_tools_batch",
    "heaac_validate_frame", "heaac_he_check_batch", "heaac_couple_after_imdct_batch",
    # heaac_fft.h
    "ff_fft_init", "ff_fft_end", "ff_fft_permute", "ff_fft_calc",
    "ff_mdct_init", "ff_mdct_end", "ff_imdct_half", "ff_imdct_calc",
    "ff_kbd_window_init", "ff_sine_window_init", "ff_init_ff_sine_windows", "ff_sine_windows",
    "av_mdct_init", "av_imdct_half", "av_imdct_calc", "av_mdct_calc", "av_mdct_end",
    "av_fft_init", "av_fft_permute", "av_fft_calc", "av_fft_end",
    # heaac_codec.h
    "heaac_aac_decoder", "heaac_codec_open", "heaac_codec_decode", "heaac_codec_close",
    # heaac_parse.h
    "heaac_asc_parse", "heaac_ga_specific_config", "heaac_aac_parse_frame_ex", "heaac_pcm_interleave_batch", "heaac_aac_layout_default", "heaac_aac_layout_from_pce", "heaac_aac_layout_from_au", "heaac_asc_layout", "heaac_aac_parse_frame_layout", "heaac_aac_parse_frame_layout_ex", "heaac_spectral_tools_batch_ex", "heaac_codec_get_context_defaults", "heaac_adts_parse_header", "heaac_adts_probe", "heaac_adts_split",
    "heaac_heaac_parse_frame_ex", "heaac_pipeline_create", "heaac_pipeline_destroy", "heaac_pipeline_submit",
    "heaac_pipeline_collect", "heaac_pipeline_timing",
    "heaac_layout_pipeline_create", "heaac_layout_pipeline_destroy", "heaac_layout_pipeline_submit",
    "heaac_layout_pipeline_collect", "heaac_layout_pipeline_channels",
    # heaac_debug.h
    "heaac_debug_workspace", "heaac_debug_xbands",
    "heaac_multi_shard", "heaac_multi_create", "heaac_multi_destroy", "heaac_multi_devices", "heaac_multi_device",
    "heaac_multi_stream", "heaac_multi_he_decode", "heaac_aac_parse_frame", "heaac_aac_parse_batch",
    "heaac_aac_tables_fingerprint",
    "heaac_sbr_table_create", "heaac_sbr_table_destroy", "heaac_sbr_table_count", "heaac_sbr_table_data",
    "heaac_sbr_stream_init", "heaac_sbr_stream_bytes", "heaac_sbr_parse_payload", "heaac_sbr_no_payload",
    "heaac_heaac_parse_frame", "heaac_heaac_parse_batch", "heaac_sbr_tables_fingerprint",
]


# heaac_dsp.h: first rule a record breaks
BAD_RULES = ["NONE", "HDR_INDEX", "HDR_RANGE", "HDR_COUNTS", "HDR_TABLE", "HDR_MAP", "HDR_FLAGS", "HDR_UNSTARTED",
             "SBR_NUM_ENV", "SBR_T_ENV", "SBR_T_Q", "SBR_FLAGS", "SBR_OLD_RANGE",
             "PS_NUM_ENV", "PS_BORDER", "PS_NR_PAR", "PS_PAR"]


class HeaacError(RuntimeError):
    pass


def validate_frame(cfg, sbr, hdr, ps=None):
    """Host-side record check of ONE frame (numpy records): returns the name of the first rule broken,
    "NONE" if the frame is valid."""
    sbr = np.ascontiguousarray(sbr, dtype=SBR_FRAME_DT).reshape(-1)[:1]
    hdr = np.ascontiguousarray(hdr, dtype=SBR_HDR_DT).reshape(-1)
    p = None
    if ps is not None:
        ps = np.ascontiguousarray(ps, dtype=PS_FRAME_DT).reshape(-1)[:1]
        p = ps.ctypes.data_as(C.c_void_p)
    r = lib().heaac_validate_frame(C.c_int(cfg), sbr.ctypes.data_as(C.c_void_p), hdr.ctypes.data_as(C.c_void_p),
                                   C.c_size_t(hdr.shape[0]), p)
    if r < 0:
        raise HeaacError("heaac_validate_frame: bad call")
    return BAD_RULES[r]


_lib = None


KERNEL_SOURCE_EXTRA = ("kernels.h", "tables.h", "validate.h")


def kernel_source_files():
    """The device sources of the kernels bench.py's workloads launch, relative to the repository root: csrc/k_*.hip,
    csrc/k_*.h, kernels.h, tables.h, validate.h and the record header include/heaac_dsp.h -- without k_tools.hip (the
    spectral tools: no bench workload runs them; tools/tools_rate.py measures them by themselves)."""
    import glob
    d = os.path.join(_HERE, "csrc")
    files = sorted(f for f in glob.glob(os.path.join(d, "k_*.hip")) + glob.glob(os.path.join(d, "k_*.h"))
                   if os.path.basename(f) != "k_tools.hip")
    files += [os.path.join(d, f) for f in KERNEL_SOURCE_EXTRA]
    files.append(os.path.join(os.path.dirname(_HERE), "include", "heaac_dsp.h"))
    return [os.path.relpath(f, os.path.dirname(_HERE)) for f in files]


def kernel_source_sha(read=None):
    """SHA-256 over the CODE of the device sources (comments and white space do not count; host-side sources --
    parsers, pipelines, the C API -- do not either): a measurement stored under profiles/ is tied to the kernels it was
    taken on, and bench.py refuses a traffic figure whose stamp differs.  read(relative path) -> text lets
    tools/traffic_stamp.py hash the files of another commit."""
    import hashlib
    import re
    root = os.path.dirname(_HERE)
    if read is None:
        read = lambda rel: open(os.path.join(root, rel), encoding="utf-8").read()
    h = hashlib.sha256()
    for rel in kernel_source_files():
        txt = read(rel)
        txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)          # block comments
        txt = re.sub(r"//[^\n]*", " ", txt)                       # line comments (no string literal of these files holds //)
        txt = " ".join(txt.split())
        h.update(os.path.basename(rel).encode())
        h.update(txt.encode())
    return h.hexdigest()[:16]


def lib():
    """Load libheaac_amd.so; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HeaacError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C ffmpeg-heaac_amd/csrc). There is no CPU fallback." % LIB_PATH)
        # PyTorch ships its own HIP runtime.  If this library is loaded first it binds /opt/rocm's copy, torch then
        # brings a second one into the process and whichever initialises later sees no device: load torch first.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        if LIB_OVERRIDDEN:
            import sys
            print("ffmpeg-heaac_amd: HEAAC_LIB_PATH is set -- using %s instead of the product library" % LIB_PATH,
                  file=sys.stderr)
        _lib = C.CDLL(LIB_PATH)
        _lib.heaac_strerror.restype = C.c_char_p
        _lib.heaac_build_info.restype = C.c_char_p
        _lib.heaac_device_workspace_bytes.restype = C.c_size_t
        _lib.heaac_get_table.argtypes = [C.c_char_p, C.c_void_p, C.c_int]
    return _lib


def _check(rc, what):
    if rc != 0:
        raise HeaacError("%s failed: %s (%d)" % (what, lib().heaac_strerror(rc).decode(), rc))


def get_table(name, n=4096):
    """Host-built table by name (for table-parity tests; no GPU needed)."""
    buf = np.zeros(n, np.float32)
    r = lib().heaac_get_table(name.encode(), buf.ctypes.data_as(C.c_void_p), n)
    if r < 0:
        raise KeyError(name)
    return buf[:r].copy()


def sbr_make_header(sample_rate=48000, start_freq=5, stop_freq=9, xover=0, freq_scale=2,
                    alter_scale=1, noise_bands=2, limiter_bands=2, limiter_gains=2,
                    interpol_freq=1, smoothing_mode=1, amp_res=1):
    """heaac_sbr_make_header(): SBR header -> band tables (host C, no GPU)."""
    h = np.zeros(1, SBR_HDR_DT)
    rc = lib().heaac_sbr_make_header(h.ctypes.data_as(C.c_void_p), sample_rate, start_freq, stop_freq,
                                     xover, freq_scale, alter_scale, noise_bands, limiter_bands,
                                     limiter_gains, interpol_freq, smoothing_mode, amp_res)
    if rc != 0:
        raise ValueError("invalid SBR header (%d)" % rc)
    return h


def _ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensor required"
    return C.c_void_p(t.data_ptr())


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def to_device(a, device="cuda"):
    """numpy (possibly structured) array -> uint8/float32 torch tensor on the GPU."""
    import torch
    a = np.ascontiguousarray(a)
    if a.dtype.names is not None or a.dtype.kind not in "fiu":
        return torch.from_numpy(a.view(np.uint8).reshape(-1)).to(device)
    return torch.from_numpy(a).to(device)


class Device:
    """HeaacDevice: per-GPU immutable tables (+ workspace)."""

    def __init__(self, max_frames=0):
        import torch
        if not torch.cuda.is_available():
            raise HeaacError("no HIP device visible: the HE-AAC DSP path has no CPU fallback")
        self._h = C.c_void_p()
        _check(lib().heaac_device_create(C.byref(self._h), C.c_size_t(max_frames)), "heaac_device_create")

    def close(self):
        if self._h:
            lib().heaac_device_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def x_bands_shares(self, n):
        """include/heaac_debug.h: the share of (frame, channel) X records of the last HE decode call that were
        stored with 32 / 48 / 64 bands, as {"32": .., "48": .., "64": ..}."""
        buf = np.zeros(2 * n, np.uint8)
        _check(lib().heaac_debug_xbands(self._h, buf.ctypes.data_as(C.c_void_p), C.c_size_t(n)), "heaac_debug_xbands")
        return {str(b): round(float((buf == b).mean()), 4) for b in (32, 48, 64)}

    # -- transforms --
    def imdct_half(self, which, x):
        import torch
        n_half = {0: 1024, 1: 128, 2: 64, 3: 64}[which]
        assert x.dtype == torch.float32 and x.shape[-1] == n_half
        x = x.contiguous()
        out = torch.empty_like(x)
        _check(lib().heaac_imdct_half_batch(self._h, which, _ptr(out), _ptr(x),
                                            C.c_size_t(x.numel() // n_half), _stream()),
               "heaac_imdct_half_batch")
        return out

    # -- AAC-LC --
    def lc_decode(self, channels, coeffs, ics, state_in, state_out=None, pcm=None, pcm_format=PCM_F32):
        import torch
        n = coeffs.shape[0]
        assert coeffs.dtype == torch.float32 and coeffs.numel() == n * channels * 1024
        assert ics.numel() == n * channels * 4 and ics.dtype == torch.uint8
        assert state_in.numel() == n * channels * 512
        if state_out is None:
            state_out = torch.empty_like(state_in)
        if pcm is None:
            if pcm_format == PCM_F32:
                pcm = torch.empty((n, channels, 1024), dtype=torch.float32, device=coeffs.device)
            else:
                pcm = torch.empty((n, 1024, channels), dtype=torch.int16, device=coeffs.device)
        _check(lib().heaac_lc_decode_batch(self._h, channels, _ptr(coeffs), _ptr(ics), _ptr(state_in),
                                           _ptr(state_out), _ptr(pcm), pcm_format, C.c_size_t(n), _stream()),
               "heaac_lc_decode_batch")
        return pcm, state_out

    # -- spectral tools before the IMDCT (M/S, intensity stereo, TNS), in place --
    def spectral_tools(self, channels, coeffs, tools, rng=None, pred=None):
        """rng: int32 [n] generator states (updated in place) -> noise substitution runs too.
        pred: float32 [n][channels][672][6] predictor states (in place) -> AAC-Main prediction too."""
        import torch
        n = coeffs.shape[0]
        assert coeffs.dtype == torch.float32 and coeffs.numel() == n * channels * 1024
        assert tools.dtype == torch.uint8 and tools.numel() == n * TOOLS_FRAME_DT.itemsize
        assert rng is None or (rng.dtype == torch.int32 and rng.numel() == n)
        assert pred is None or (pred.dtype == torch.float32 and pred.numel() == n * channels * MAX_PREDICTORS * 6)
        _check(lib().heaac_spectral_tools_batch(self._h, channels, _ptr(coeffs), _ptr(tools),
                                                _ptr(rng) if rng is not None else None,
                                                _ptr(rng) if rng is not None else None,
                                                _ptr(pred) if pred is not None else None,
                                                _ptr(pred) if pred is not None else None,
                                                C.c_size_t(n), _stream()), "heaac_spectral_tools_batch")
        return coeffs

    def spectral_tools_ex(self, channels, stages, coeffs, tools, rng=None, pred=None, cce=None, cce_coeffs=None):
        """heaac_spectral_tools_batch_ex: the PRE / POST halves; cce [n][n_cce] records (uint8 tensor) and
        cce_coeffs [n][n_cce][1024] couple into the target in POST."""
        import torch
        n = coeffs.shape[0]
        assert coeffs.dtype == torch.float32 and coeffs.numel() == n * channels * 1024
        assert tools.dtype == torch.uint8 and tools.numel() == n * TOOLS_FRAME_DT.itemsize
        n_cce = 0
        if cce is not None:
            n_cce = cce.numel() // (n * CCE_FRAME_DT.itemsize)
            assert cce.dtype == torch.uint8 and cce.numel() == n * n_cce * CCE_FRAME_DT.itemsize
            assert cce_coeffs.dtype == torch.float32 and cce_coeffs.numel() == n * n_cce * 1024
        _check(lib().heaac_spectral_tools_batch_ex(self._h, channels, stages, _ptr(coeffs), _ptr(tools),
                                                   _ptr(rng) if rng is not None else None,
                                                   _ptr(rng) if rng is not None else None,
                                                   _ptr(pred) if pred is not None else None,
                                                   _ptr(pred) if pred is not None else None,
                                                   _ptr(cce) if n_cce else None, _ptr(cce_coeffs) if n_cce else None,
                                                   n_cce, C.c_size_t(n), _stream()), "heaac_spectral_tools_batch_ex")
        return coeffs

    # -- HE-AAC --
    def couple_after_imdct(self, channels, pcm, cce, coupling, s16=False):
        """heaac_couple_after_imdct_batch: pcm [n][channels][1024] f32 updated in place; returns the int16
        interleave of the result when s16."""
        import torch
        n = pcm.shape[0]
        assert pcm.dtype == torch.float32 and pcm.numel() == n * channels * 1024, "pcm"
        assert cce.dtype == torch.float32 and cce.numel() == n * 1024, "cce"
        assert coupling.numel() * coupling.element_size() == n * COUPLING_DT.itemsize, "coupling"
        out = torch.empty((n, 1024, channels), dtype=torch.int16, device=pcm.device) if s16 else None
        _check(lib().heaac_couple_after_imdct_batch(self._h, channels, _ptr(pcm), _ptr(cce), _ptr(coupling),
                                                    _ptr(out), C.c_size_t(n), _stream()),
               "heaac_couple_after_imdct_batch")
        return out

    def pcm_interleave(self, planes, length, pcm_format=None):
        """heaac_pcm_interleave_batch.  planes: one (tensor, element offset, frame stride in floats) per output channel,
        each tensor float32 on the device holding that channel of all n frames; returns int16 [n][length][channels]."""
        import torch
        fmt = PCM_S16 if pcm_format is None else pcm_format
        ch = len(planes)
        t0, off0, stride0 = planes[0]
        n = (t0.numel() - off0 - length) // stride0 + 1 if stride0 else 1
        refs = (_PlaneRef * ch)()
        for c, (t, off, stride) in enumerate(planes):
            assert t.dtype == torch.float32 and off + (n - 1) * stride + length <= t.numel(), "plane %d" % c
            refs[c].d_base = t.data_ptr() + 4 * off
            refs[c].frame_stride = stride
        out = torch.empty((n, length, ch), dtype=torch.int16, device=t0.device)
        _check(lib().heaac_pcm_interleave_batch(self._h, ch, refs, int(length), int(fmt), _ptr(out), C.c_size_t(n), _stream()),
               "heaac_pcm_interleave_batch")
        return out

    def he_check(self, cfg, sbr, hdr, ps=None):
        """heaac_he_check_batch on device-resident records (byte tensors as he_decode takes them):
        returns None if every frame is valid, else (first bad frame index, rule name)."""
        import torch
        n = sbr.numel() // SBR_FRAME_DT.itemsize
        first, rule = C.c_size_t(0), C.c_int(0)
        rc = lib().heaac_he_check_batch(self._h, C.c_int(cfg), C.c_void_p(sbr.data_ptr()), C.c_void_p(hdr.data_ptr()),
                                        C.c_size_t(hdr.numel() // SBR_HDR_DT.itemsize),
                                        C.c_void_p(ps.data_ptr()) if ps is not None else None, C.c_size_t(n),
                                        C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(first), C.byref(rule))
        if rc == 0:
            return None
        if rc != -1:
            _check(rc, "heaac_he_check_batch")
        if first.value == C.c_size_t(-1).value:
            raise HeaacError("heaac_he_check_batch: bad arguments")
        return int(first.value), BAD_RULES[rule.value] if 0 <= rule.value < len(BAD_RULES) else str(rule.value)

    def he_decode(self, cfg, coeffs, ics, sbr, hdr, ps, state_in, state_out=None, pcm=None,
                  pcm_format=PCM_F32, downsampled=False):
        """downsampled: HEAAC_HE_DOWNSAMPLED -- output at the core rate, 1024 samples per channel."""
        import torch
        n = coeffs.shape[0]
        # a wrong-length tensor would be a silent device out-of-bounds read: check them all here
        assert coeffs.dtype == torch.float32 and coeffs.numel() == n * CORE_CH[cfg] * 1024, "coeffs"
        assert state_in.dtype == torch.float32 and state_in.numel() == n * STATE_WORDS[cfg], "state_in"
        assert ics.numel() * ics.element_size() == n * CORE_CH[cfg] * ICS_DT.itemsize, "ics"
        assert sbr.numel() * sbr.element_size() == n * SBR_FRAME_DT.itemsize, "sbr"
        assert hdr.numel() * hdr.element_size() >= SBR_HDR_DT.itemsize and \
            (hdr.numel() * hdr.element_size()) % SBR_HDR_DT.itemsize == 0, "hdr"
        if cfg == CFG_HEV2:
            assert ps is not None and ps.numel() * ps.element_size() == n * PS_FRAME_DT.itemsize, "ps"
        if state_out is None:
            state_out = torch.empty_like(state_in)
        else:
            assert state_out.dtype == torch.float32 and state_out.numel() == state_in.numel(), "state_out"
        length = 1024 if downsampled else 2048
        if pcm is None:
            if pcm_format == PCM_F32:
                pcm = torch.empty((n, OUT_CH[cfg], length), dtype=torch.float32, device=coeffs.device)
            else:
                pcm = torch.empty((n, length, OUT_CH[cfg]), dtype=torch.int16, device=coeffs.device)
        else:
            assert pcm.numel() == n * OUT_CH[cfg] * length, "pcm"
        n_hdr = hdr.numel() // SBR_HDR_DT.itemsize
        _check(lib().heaac_he_decode_batch_ex(self._h, cfg, 1 if downsampled else 0, _ptr(coeffs), _ptr(ics), _ptr(sbr),
                                              _ptr(hdr), C.c_size_t(n_hdr), _ptr(ps), _ptr(state_in), _ptr(state_out),
                                              _ptr(pcm), pcm_format, C.c_size_t(n), _stream()),
               "heaac_he_decode_batch")
        return pcm, state_out

    def qmf_analysis(self, x, xhist, scale=32768.0):
        import torch
        n = x.shape[0]
        W = torch.empty((n, 32, 32, 2), dtype=torch.float32, device=x.device)
        xh = torch.empty_like(xhist)
        _check(lib().heaac_qmf_analysis_batch(self._h, _ptr(x), _ptr(xhist), _ptr(xh), _ptr(W),
                                              C.c_float(scale), C.c_size_t(n), _stream()),
               "heaac_qmf_analysis_batch")
        return W, xh

    def qmf_synthesis(self, X, v, scale=2.0 ** -15, bias=385.0):
        import torch
        n = X.shape[0]
        out = torch.empty((n, 2048), dtype=torch.float32, device=X.device)
        vo = torch.empty_like(v)
        _check(lib().heaac_qmf_synthesis_batch(self._h, _ptr(X), _ptr(v), _ptr(vo), _ptr(out),
                                               C.c_float(scale), C.c_float(bias), C.c_size_t(n), _stream()),
               "heaac_qmf_synthesis_batch")
        return out, vo

    def qmf_synthesis_ds(self, X, v, scale=2.0 ** -15, bias=385.0):
        """Downsampled synthesis bank (div = 1): X [n][2][32][64], v [n][576] -> out [n][1024]."""
        import torch
        n = X.shape[0]
        out = torch.empty((n, 1024), dtype=torch.float32, device=X.device)
        vo = torch.empty_like(v)
        _check(lib().heaac_qmf_synthesis_ds_batch(self._h, _ptr(X), _ptr(v), _ptr(vo), _ptr(out),
                                                  C.c_float(scale), C.c_float(bias), C.c_size_t(n), _stream()),
               "heaac_qmf_synthesis_ds_batch")
        return out, vo


# ---------------------------------------------------------------------------------------------------
# host-side AAC parser (include/heaac_parse.h)
# ---------------------------------------------------------------------------------------------------
class _PlaneRef(C.Structure):
    _fields_ = [("d_base", C.c_void_p), ("frame_stride", C.c_size_t)]


class AacConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("object_type", "sampling_index", "sample_rate", "chan_config", "sbr",
                                       "ext_object_type", "ext_sampling_index", "ext_sample_rate",
                                       "ext_chan_config", "ps")]


class AdtsHeader(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("sample_rate", "samples", "bit_rate", "object_type", "sampling_index",
                                       "chan_config", "crc_absent", "num_aac_frames", "frame_length")]


COUPLING_DT = np.dtype([("gain", "<f4", (2,)), ("on", "u1", (2,)), ("pad", "u1", (2,))])
AAC_STREAM_DT = np.dtype([("window_sequence", "u1", (2,)), ("use_kb_window", "u1", (2,)),
                          ("cce_window_sequence", "u1", (16,)), ("cce_use_kb_window", "u1", (16,)),
                          ("mapped_tag", "u1"), ("pad", "u1", (3,))])
AAC_INFO_DT = np.dtype([("channels", "<i4"), ("bits_consumed", "<i4"), ("sbr_payload_bit", "<i4"),
                        ("sbr_payload_bytes", "<i4"), ("sbr_crc", "<i4"), ("elem_id", "<i4"), ("n_cce", "<i4"),
                        ("sbr_misplaced", "<i4"), ("refused", "<i4")])
REFUSED_AS_REFERENCE, REFUSED_RUN_TOOLS = 1, 2


def asc_parse(buf):
    """heaac_asc_parse: (AacConfig, bit offset of the specific config)."""
    c = AacConfig()
    r = lib().heaac_asc_parse(C.byref(c), bytes(buf), len(buf))
    if r < 0:
        raise HeaacError("heaac_asc_parse -> %d" % r)
    return c, r


def adts_parse_header(buf):
    h = AdtsHeader()
    r = lib().heaac_adts_parse_header(C.byref(h), bytes(buf), len(buf))
    return h, r


ADTS_PACKET_DT = np.dtype([("offset", "<u8"), ("size", "<u8"), ("kind", "<i4"), ("header_size", "<i4")])
ADTS_FRAME, ADTS_JUNK, ADTS_TRUNCATED, ADTS_TAG = 0, 1, 2, 3


def adts_probe(buf):
    """heaac_adts_probe: the demuxer's probe score of a raw ADTS buffer."""
    return lib().heaac_adts_probe(bytes(buf), C.c_size_t(len(buf)))


def adts_split(buf):
    """heaac_adts_split: (packets [ADTS_PACKET_DT], header of the first frame or None)."""
    buf = bytes(buf)
    L = lib()
    L.heaac_adts_split.restype = C.c_long
    n = L.heaac_adts_split(buf, C.c_size_t(len(buf)), None, C.c_size_t(0), None)
    if n < 0:
        raise HeaacError("heaac_adts_split -> %d" % n)
    out = np.zeros(n, ADTS_PACKET_DT)
    h = AdtsHeader()
    m = L.heaac_adts_split(buf, C.c_size_t(len(buf)), out.ctypes.data_as(C.c_void_p), C.c_size_t(n), C.byref(h))
    assert m == n
    return out, (h if (out["kind"] != ADTS_JUNK).any() and ((out["kind"] == ADTS_FRAME) | (out["kind"] == ADTS_TRUNCATED)).any() else None)


class Pipeline:
    """include/heaac_pipeline.h: access units in host memory -> int16 PCM in host memory, ticks overlapped."""

    def __init__(self, aac_cfg, he_cfg, n_streams, threads=0):
        self._h = C.c_void_p()
        self.n, self.ch, self.len = n_streams, OUT_CH[he_cfg], OUT_LEN[he_cfg]
        if self.len == 2048 and aac_cfg.ext_sample_rate and aac_cfg.ext_sample_rate < 2 * aac_cfg.sample_rate:
            self.len = 1024                                # downsampled SBR: the output at the core rate
        _check(lib().heaac_pipeline_create(C.byref(self._h), C.byref(aac_cfg), he_cfg, C.c_size_t(n_streams), threads),
               "heaac_pipeline_create")

    def submit(self, aus, with_status=True):
        """aus: n_streams access units (bytes).  Returns the parse status per stream (None without a status array)."""
        assert len(aus) == self.n
        keep = [C.create_string_buffer(bytes(a), len(a)) for a in aus]
        ptrs = (C.c_char_p * self.n)(*[C.cast(k, C.c_char_p) for k in keep])
        sizes = (C.c_int * self.n)(*[len(a) for a in aus])
        status = np.zeros(self.n, np.int32) if with_status else None
        _check(lib().heaac_pipeline_submit(self._h, ptrs, sizes, status.ctypes.data_as(C.c_void_p) if with_status else None),
               "heaac_pipeline_submit")
        return status

    def submit_raw(self, ptrs, sizes):
        _check(lib().heaac_pipeline_submit(self._h, ptrs, sizes, None), "heaac_pipeline_submit")

    def collect(self):
        """PCM of the oldest tick in flight: int16 [n][2048][channels] (a view of the pipeline's pinned buffer)."""
        p = C.POINTER(C.c_int16)()
        _check(lib().heaac_pipeline_collect(self._h, C.byref(p)), "heaac_pipeline_collect")
        return np.ctypeslib.as_array(p, shape=(self.n, self.len, self.ch))

    def timing(self):
        ms = (C.c_float * 4)()
        lib().heaac_pipeline_timing(self._h, ms)
        return dict(parse=ms[0], h2d=ms[1], gpu=ms[2], d2h=ms[3])

    def close(self):
        if self._h:
            lib().heaac_pipeline_destroy(self._h)
            self._h = C.c_void_p()


class LayoutPipeline:
    """include/heaac_pipeline.h, second half: n streams of one multi-element layout, access units in, int16 PCM out."""

    def __init__(self, aac_cfg, layout, n_streams, threads=0):
        self._h = C.c_void_p()
        self.n, self.ch = n_streams, int(layout[0]["channels"])
        he = aac_cfg.sbr == 1
        self.len = 2048 if he and not (aac_cfg.ext_sample_rate and aac_cfg.ext_sample_rate < 2 * aac_cfg.sample_rate) else 1024
        self._layout = np.ascontiguousarray(layout)
        _check(lib().heaac_layout_pipeline_create(C.byref(self._h), C.byref(aac_cfg), self._layout.ctypes.data_as(C.c_void_p),
                                                  C.c_size_t(n_streams), threads), "heaac_layout_pipeline_create")
        self.ch = int(lib().heaac_layout_pipeline_channels(self._h))

    def submit(self, aus):
        assert len(aus) == self.n
        keep = [C.create_string_buffer(bytes(a), len(a)) for a in aus]
        ptrs = (C.c_char_p * self.n)(*[C.cast(k, C.c_char_p) for k in keep])
        sizes = (C.c_int * self.n)(*[len(a) for a in aus])
        status = np.zeros(self.n, np.int32)
        _check(lib().heaac_layout_pipeline_submit(self._h, ptrs, sizes, status.ctypes.data_as(C.c_void_p)),
               "heaac_layout_pipeline_submit")
        return status

    def collect(self):
        p = C.POINTER(C.c_int16)()
        _check(lib().heaac_layout_pipeline_collect(self._h, C.byref(p)), "heaac_layout_pipeline_collect")
        return np.ctypeslib.as_array(p, shape=(self.n, self.len, self.ch))

    def close(self):
        if self._h:
            lib().heaac_layout_pipeline_destroy(self._h)
            self._h = C.c_void_p()


class _CceOut(C.Structure):
    _fields_ = [("cce", C.c_void_p), ("coeffs", C.c_void_p), ("ics", C.c_void_p), ("tools", C.c_void_p), ("elem", C.c_void_p)]


def aac_parse_frame_ex(cfg, stream, au, coeff_channels=2, with_cce=True):
    """heaac_aac_parse_frame_ex on one access unit.  `stream`: one AAC_STREAM_DT record (updated in place).
    Returns (status, dict(coeffs [coeff_channels][1024], ics, tools, info, cce [MAX_CCE], cce_coeffs, cce_ics, cce_tools))."""
    au = bytes(au)
    out = dict(coeffs=np.zeros((coeff_channels, 1024), np.float32), ics=np.zeros(2, ICS_DT),
               tools=np.zeros(1, TOOLS_FRAME_DT), info=np.zeros(1, AAC_INFO_DT),
               cce=np.zeros(MAX_CCE, CCE_FRAME_DT), cce_coeffs=np.zeros((MAX_CCE, 1024), np.float32),
               cce_ics=np.zeros(MAX_CCE, ICS_DT), cce_tools=np.zeros(MAX_CCE, TOOLS_FRAME_DT))
    co = _CceOut(out["cce"].ctypes.data, out["cce_coeffs"].ctypes.data, out["cce_ics"].ctypes.data,
                 out["cce_tools"].ctypes.data)
    r = lib().heaac_aac_parse_frame_ex(C.byref(cfg), stream.ctypes.data_as(C.c_void_p), au, len(au), coeff_channels,
                                       out["coeffs"].ctypes.data_as(C.c_void_p), out["ics"].ctypes.data_as(C.c_void_p),
                                       out["tools"].ctypes.data_as(C.c_void_p), C.byref(co) if with_cce else None,
                                       out["info"].ctypes.data_as(C.c_void_p))
    return r, out


# ---- channel layouts (several output elements per access unit) ----
MAX_ELEMENTS = 16
AAC_ELEM_SLOT_DT = np.dtype([("type", "u1"), ("id", "u1"), ("channels", "u1"), ("first_channel", "u1")])
AAC_LAYOUT_DT = np.dtype([("chan_config", "<i4"), ("n_elements", "<i4"), ("channels", "<i4"), ("tags_mapped", "<i4"),
                          ("channel_layout", "<i8"), ("elem", AAC_ELEM_SLOT_DT, (MAX_ELEMENTS,)),
                          ("slot_of", "i1", (4, 16)), ("tag_map", "i1", (4, 16))])
AAC_ELEM_INFO_DT = np.dtype([("present", "u1"), ("type", "u1"), ("tag", "u1"), ("seq", "u1"), ("sbr_crc", "u1"),
                             ("sbr_misplaced", "u1"), ("pad", "u1", (2,)), ("sbr_payload_bit", "<i4"), ("sbr_payload_bytes", "<i4")])
assert AAC_LAYOUT_DT.itemsize == 216 and AAC_ELEM_INFO_DT.itemsize == 16


def aac_layout_default(chan_config):
    """heaac_aac_layout_default: (status, one AAC_LAYOUT_DT record)."""
    l = np.zeros(1, AAC_LAYOUT_DT)
    return lib().heaac_aac_layout_default(l.ctypes.data_as(C.c_void_p), int(chan_config)), l


def aac_layout_from_pce(buf, bit_offset):
    """heaac_aac_layout_from_pce: (status, layout, bits used)."""
    l = np.zeros(1, AAC_LAYOUT_DT)
    used = C.c_int(0)
    buf = bytes(buf)
    r = lib().heaac_aac_layout_from_pce(l.ctypes.data_as(C.c_void_p), buf, len(buf), int(bit_offset), C.byref(used))
    return r, l, used.value


def aac_layout_from_au(au):
    """heaac_aac_layout_from_au: (status, layout)."""
    l = np.zeros(1, AAC_LAYOUT_DT)
    au = bytes(au)
    return lib().heaac_aac_layout_from_au(l.ctypes.data_as(C.c_void_p), au, len(au)), l


def asc_layout(buf):
    """heaac_asc_layout: (status, AacConfig, layout)."""
    c = AacConfig()
    l = np.zeros(1, AAC_LAYOUT_DT)
    buf = bytes(buf)
    return lib().heaac_asc_layout(C.byref(c), l.ctypes.data_as(C.c_void_p), buf, len(buf)), c, l


def aac_parse_frame_layout(cfg, layout, streams, au, with_cce=False):
    """heaac_aac_parse_frame_layout (with_cce: heaac_aac_parse_frame_layout_ex) on one access unit.  layout: one
    AAC_LAYOUT_DT record (its tag map is updated), streams: AAC_STREAM_DT [n_elements] (updated).  Returns (status,
    dict(coeffs [ne][2][1024], ics [ne][2], tools [ne], elem [ne], info; with_cce: cce [ne][MAX_CCE], cce_coeffs
    [MAX_CCE][1024], cce_ics [MAX_CCE], cce_tools [MAX_CCE], cce_elem [MAX_CCE] (not handed over when
    with_cce == "no_sbr")))."""
    au = bytes(au)
    ne = int(layout[0]["n_elements"])
    assert streams.dtype == AAC_STREAM_DT and streams.shape[0] >= ne
    out = dict(coeffs=np.zeros((ne, 2, 1024), np.float32), ics=np.zeros((ne, 2), ICS_DT), tools=np.zeros(ne, TOOLS_FRAME_DT),
               elem=np.zeros(ne, AAC_ELEM_INFO_DT), info=np.zeros(1, AAC_INFO_DT))
    args = (C.byref(cfg), layout.ctypes.data_as(C.c_void_p), streams.ctypes.data_as(C.c_void_p),
            au, len(au), out["coeffs"].ctypes.data_as(C.c_void_p),
            out["ics"].ctypes.data_as(C.c_void_p), out["tools"].ctypes.data_as(C.c_void_p),
            out["elem"].ctypes.data_as(C.c_void_p))
    if not with_cce:
        return lib().heaac_aac_parse_frame_layout(*args, out["info"].ctypes.data_as(C.c_void_p)), out
    out.update(cce=np.zeros((ne, MAX_CCE), CCE_FRAME_DT), cce_coeffs=np.zeros((MAX_CCE, 1024), np.float32),
               cce_ics=np.zeros(MAX_CCE, ICS_DT), cce_tools=np.zeros(MAX_CCE, TOOLS_FRAME_DT),
               cce_elem=np.zeros(MAX_CCE, AAC_ELEM_INFO_DT))
    co = _CceOut(out["cce"].ctypes.data, out["cce_coeffs"].ctypes.data, out["cce_ics"].ctypes.data,
                 out["cce_tools"].ctypes.data, out["cce_elem"].ctypes.data if with_cce != "no_sbr" else None)
    return lib().heaac_aac_parse_frame_layout_ex(*args, C.byref(co), out["info"].ctypes.data_as(C.c_void_p)), out


def aac_parse_batch(cfg, streams, aus, threads=0):
    """heaac_aac_parse_batch over a list of access units (bytes), one per stream.  `streams`: AAC_STREAM_DT
    array updated in place.  Returns dict(coeffs [n][2][1024], ics [n][2], tools [n], info [n], status [n])."""
    n = len(aus)
    keep = [C.create_string_buffer(bytes(a), len(a)) for a in aus]
    ptrs = (C.c_char_p * n)(*[C.cast(k, C.c_char_p) for k in keep])
    sizes = (C.c_int * n)(*[len(a) for a in aus])
    out = dict(coeffs=np.zeros((n, 2, 1024), np.float32), ics=np.zeros((n, 2), ICS_DT),
               tools=np.zeros(n, TOOLS_FRAME_DT), info=np.zeros(n, AAC_INFO_DT), status=np.zeros(n, np.int32))
    assert streams.dtype == AAC_STREAM_DT and streams.shape == (n,)
    failed = lib().heaac_aac_parse_batch(C.byref(cfg), streams.ctypes.data_as(C.c_void_p), ptrs, sizes, C.c_size_t(n),
                                         out["coeffs"].ctypes.data_as(C.c_void_p), out["ics"].ctypes.data_as(C.c_void_p),
                                         out["tools"].ctypes.data_as(C.c_void_p), out["info"].ctypes.data_as(C.c_void_p),
                                         out["status"].ctypes.data_as(C.c_void_p), C.c_int(threads))
    out["failed"] = failed
    return out


# ---------------------------------------------------------------------------------------------
# heaac_parse.h, second slice: SBR / PS payloads
# ---------------------------------------------------------------------------------------------
SBR_PARSE_INFO_DT = np.dtype([("sbr_bits", "<i4"), ("header", "<i4"), ("ps_present", "<i4"), ("ps_status", "<i4")])
PARSE_NO_SBR = 1


class SbrHeaderTable:
    """heaac_sbr_table_*: the batch's table of derived SBR headers (entry 0 = the null header)."""

    def __init__(self, capacity=256):
        L = lib()
        L.heaac_sbr_table_create.restype = C.c_void_p
        L.heaac_sbr_table_count.restype = C.c_size_t
        L.heaac_sbr_table_data.restype = C.c_void_p
        self._h = L.heaac_sbr_table_create(C.c_size_t(capacity))
        if not self._h:
            raise HeaacError("heaac_sbr_table_create failed")

    def __len__(self):
        return int(lib().heaac_sbr_table_count(C.c_void_p(self._h)))

    def headers(self):
        """Copy of the entries so far as an SBR_HDR_DT array (what he_decode takes as `hdr`)."""
        n = len(self)
        addr = lib().heaac_sbr_table_data(C.c_void_p(self._h))
        buf = (C.c_uint8 * (n * SBR_HDR_DT.itemsize)).from_address(addr)
        return np.frombuffer(bytes(buf), dtype=SBR_HDR_DT).copy()

    def close(self):
        if self._h:
            lib().heaac_sbr_table_destroy(C.c_void_p(self._h))
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sbr_streams(n):
    """n fresh HeaacSbrStream records (opaque bytes, heaac_sbr_stream_init)."""
    L = lib()
    L.heaac_sbr_stream_bytes.restype = C.c_size_t
    st = np.zeros((n, int(L.heaac_sbr_stream_bytes())), np.uint8)
    L.heaac_sbr_stream_init(st.ctypes.data_as(C.c_void_p), C.c_size_t(n))
    return st


def sbr_parse_payload(stream, table, sample_rate, payload, channels, allow_ps, crc=False, bit=0, cnt=None, misplaced=False):
    """heaac_sbr_parse_payload on ONE stream record (a row of sbr_streams()).  `payload`: the bytes that
    follow the 4-bit extension type when bit = 0 (tests), or a whole access unit with `bit` set.
    Returns (status, sbr record, ps record, info)."""
    assert stream.dtype == np.uint8 and stream.flags["C_CONTIGUOUS"]
    payload = bytes(payload)
    sbr = np.zeros(1, SBR_FRAME_DT)
    ps = np.zeros(1, PS_FRAME_DT)
    info = np.zeros(1, SBR_PARSE_INFO_DT)
    r = lib().heaac_sbr_parse_payload(stream.ctypes.data_as(C.c_void_p), C.c_void_p(table._h), C.c_int(sample_rate),
                                      payload, C.c_int(len(payload)), C.c_int(bit),
                                      C.c_int(len(payload) if cnt is None else cnt), C.c_int(bool(crc)),
                                      C.c_int(channels), C.c_int(int(bool(allow_ps)) | (2 if misplaced else 0)),
                                      sbr.ctypes.data_as(C.c_void_p), ps.ctypes.data_as(C.c_void_p),
                                      info.ctypes.data_as(C.c_void_p))
    return r, sbr, ps, info[0]


def sbr_no_payload(stream, channels):
    """heaac_sbr_no_payload: the record of an access unit without an SBR payload for ONE stream record."""
    sbr = np.zeros(1, SBR_FRAME_DT)
    lib().heaac_sbr_no_payload(stream.ctypes.data_as(C.c_void_p), C.c_int(channels), sbr.ctypes.data_as(C.c_void_p), None)
    return sbr


def heaac_parse_batch(cfg, streams, sbr_st, table, aus, threads=0, with_ps=False):
    """heaac_heaac_parse_batch: whole HE-AAC access units, one per stream.  streams: AAC_STREAM_DT [n],
    sbr_st: sbr_streams(n); both updated in place.  Returns the dict of aac_parse_batch plus sbr [n], ps [n]."""
    n = len(aus)
    keep = [C.create_string_buffer(bytes(a), len(a)) for a in aus]
    ptrs = (C.c_char_p * n)(*[C.cast(k, C.c_char_p) for k in keep])
    sizes = (C.c_int * n)(*[len(a) for a in aus])
    out = dict(coeffs=np.zeros((n, 2, 1024), np.float32), ics=np.zeros((n, 2), ICS_DT),
               tools=np.zeros(n, TOOLS_FRAME_DT), info=np.zeros(n, AAC_INFO_DT), status=np.zeros(n, np.int32),
               sbr=np.zeros(n, SBR_FRAME_DT), ps=np.zeros(n, PS_FRAME_DT))
    assert streams.dtype == AAC_STREAM_DT and streams.shape == (n,) and sbr_st.shape[0] == n
    failed = lib().heaac_heaac_parse_batch(
        C.byref(cfg), streams.ctypes.data_as(C.c_void_p), sbr_st.ctypes.data_as(C.c_void_p), C.c_void_p(table._h),
        ptrs, sizes, C.c_size_t(n),
        out["coeffs"].ctypes.data_as(C.c_void_p), out["ics"].ctypes.data_as(C.c_void_p),
        out["tools"].ctypes.data_as(C.c_void_p), out["sbr"].ctypes.data_as(C.c_void_p),
        out["ps"].ctypes.data_as(C.c_void_p) if with_ps else None,
        out["info"].ctypes.data_as(C.c_void_p), out["status"].ctypes.data_as(C.c_void_p), C.c_int(threads))
    out["failed"] = failed
    return out
