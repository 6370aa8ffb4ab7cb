"""GPU: the reference-shaped per-call surfaces (FFTContext / ff_imdct_half, AVCodec-shaped
decoder) run the HIP path and agree with the oracle bit for bit."""
import ctypes as C
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class FFTComplex(C.Structure):
    _fields_ = [("re", C.c_float), ("im", C.c_float)]


class FFTContext(C.Structure):
    pass


FN2 = C.CFUNCTYPE(None, C.POINTER(FFTContext), C.c_void_p)
FN3 = C.CFUNCTYPE(None, C.POINTER(FFTContext), C.c_void_p, C.c_void_p)
FFTContext._fields_ = [
    ("nbits", C.c_int), ("inverse", C.c_int), ("revtab", C.POINTER(C.c_uint16)),
    ("exptab", C.c_void_p), ("exptab1", C.c_void_p), ("tmp_buf", C.c_void_p),
    ("mdct_size", C.c_int), ("mdct_bits", C.c_int), ("tcos", C.POINTER(C.c_float)),
    ("tsin", C.POINTER(C.c_float)), ("fft_permute", FN2), ("fft_calc", FN2),
    ("imdct_calc", FN3), ("imdct_half", FN3), ("mdct_calc", FN3),
    ("split_radix", C.c_int), ("permutation", C.c_int),
]


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("which,nbits,scale", [(0, 11, 1.0), (1, 8, 1.0), (2, 7, 1.0 / 64), (3, 7, -2.0)])
def test_ff_mdct_init_and_imdct(pkg, oracle, dev, which, nbits, scale):
    lib = pkg.lib()
    s = FFTContext()
    assert lib.ff_mdct_init(C.byref(s), nbits, 1, C.c_double(scale)) == 0
    n = 1 << nbits
    assert (s.mdct_bits, s.mdct_size, s.nbits, s.inverse, s.split_radix, s.permutation) == (nbits, n, nbits - 2, 1, 1, 0)
    name = {0: "tcos2048", 1: "tcos256", 2: "tcos128s", 3: "tcos128a"}[which]
    tc = np.ctypeslib.as_array(s.tcos, (n // 2,))
    assert np.array_equal(_bits(tc), _bits(oracle.get_table(name)))
    assert C.addressof(s.tsin.contents) - C.addressof(s.tcos.contents) == n  # tsin = tcos + n/4 floats
    rev = np.ctypeslib.as_array(s.revtab, (n // 4,))
    assert np.array_equal(rev, oracle.get_table("revtab%d" % min(which, 2)).astype(np.uint16))
    rng = np.random.default_rng(which)
    x = rng.standard_normal(n // 2).astype(np.float32)
    out = np.zeros(n // 2, np.float32)
    lib.ff_imdct_half(C.byref(s), out.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p))
    assert np.array_equal(_bits(out), _bits(oracle.imdct_half(which, x)))
    full = np.zeros(n, np.float32)
    s.imdct_calc(C.byref(s), full.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p))   # via the fn pointer
    assert np.array_equal(_bits(full), _bits(oracle.imdct_calc(which, x)))
    lib.ff_mdct_end(C.byref(s))
    assert not s.tcos and not s.revtab


def test_ff_mdct_init_rejects_foreign_transforms(pkg, dev):
    s = FFTContext()
    assert pkg.lib().ff_mdct_init(C.byref(s), 9, 1, C.c_double(1.0)) == -1
    assert pkg.lib().ff_mdct_init(C.byref(s), 11, 0, C.c_double(1.0)) == -1


@pytest.mark.parametrize("nbits", [5, 6, 9])
def test_ff_fft_permute_calc(pkg, oracle, dev, nbits):
    lib = pkg.lib()
    s = FFTContext()
    assert lib.ff_fft_init(C.byref(s), nbits, 1) == 0
    n = 1 << nbits
    rng = np.random.default_rng(nbits)
    z = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    buf = z.copy()
    lib.ff_fft_permute(C.byref(s), buf.ctypes.data_as(C.c_void_p))
    rev = oracle.get_table({9: "revtab0", 6: "revtab1", 5: "revtab2"}[nbits]).astype(int)
    zp = np.zeros_like(z); zp[rev] = z
    assert np.array_equal(buf, zp)
    lib.ff_fft_calc(C.byref(s), buf.ctypes.data_as(C.c_void_p))
    ref = oracle.fft_calc(nbits, zp)
    assert np.array_equal(buf.view(np.uint32), ref.view(np.uint32))
    lib.ff_fft_end(C.byref(s))


class HeaacPacket(C.Structure):
    """AVPacket (avcodec.h:960-1002)"""
    _fields_ = [("pts", C.c_int64), ("dts", C.c_int64), ("data", C.c_void_p), ("size", C.c_int),
                ("stream_index", C.c_int), ("flags", C.c_int), ("duration", C.c_int), ("destruct", C.c_void_p),
                ("priv", C.c_void_p), ("pos", C.c_int64), ("convergence_duration", C.c_int64)]

    def __init__(self, data=None, size=0):
        super().__init__(pts=-2 ** 63, dts=-2 ** 63, data=data, size=size, pos=-1)


class HeaacCodecContext(C.Structure):
    """AVCodecContext of libavcodec 52.78 (include/heaac_codec.h): named where the decoder reads or writes."""
    _fields_ = [("av_class", C.c_void_p), ("bit_rate", C.c_int), ("bit_rate_tolerance", C.c_int), ("flags", C.c_int),
                ("sub_id", C.c_int), ("me_method", C.c_int), ("extradata", C.c_char_p), ("extradata_size", C.c_int),
                ("opaque_44", C.c_int * 7), ("draw_horiz_band", C.c_void_p), ("sample_rate", C.c_int),
                ("channels", C.c_int), ("sample_fmt", C.c_int), ("frame_size", C.c_int), ("frame_number", C.c_int),
                ("opaque_100", C.c_int * 13), ("codec", C.c_void_p), ("priv_data", C.c_void_p),
                ("opaque_168", C.c_int * 24), ("codec_type", C.c_int), ("codec_id", C.c_int),
                ("opaque_272", C.c_int * 176), ("channel_layout", C.c_int64), ("request_channel_layout", C.c_int64),
                ("opaque_992", C.c_int * 24)]

    def __init__(self, cfg=-1, extradata=None, extradata_size=0):
        # cfg >= 0: parser-output packets of that configuration (HEAAC_SUBID_RECORDS); -1: AAC access units
        super().__init__(codec_type=-1, sub_id=(0x48450000 | cfg) if cfg >= 0 else 0, extradata=extradata,
                         extradata_size=extradata_size)


assert C.sizeof(HeaacPacket) == 72 and C.sizeof(HeaacCodecContext) == 1088
assert HeaacCodecContext.priv_data.offset == 160 and HeaacCodecContext.channel_layout.offset == 976


@pytest.mark.parametrize("cfgname", ["CFG_LC_STEREO", "CFG_HEV1", "CFG_HEV2"])
def test_codec_surface_decodes_a_stream(pkg, oracle, dev, cfgname):
    """open / decode x4 / close through the AVCodec-shaped vtable; int16 PCM equals the oracle's."""
    lib = pkg.lib()
    synth = importlib.import_module("ffmpeg_heaac_amd.synth")
    cfg = getattr(pkg, cfgname)
    rng = np.random.default_rng(77)
    ctx = HeaacCodecContext(cfg=cfg)
    codec = C.c_void_p.in_dll(lib, "heaac_aac_decoder")
    assert lib.heaac_codec_open(C.byref(ctx), C.c_void_p(C.addressof(codec))) == 0
    assert (ctx.channels, ctx.frame_size) == (pkg.OUT_CH[cfg], pkg.OUT_LEN[cfg])
    hdr = synth.default_headers(pkg)
    state = np.zeros((1, pkg.STATE_WORDS[cfg]), np.float32)
    if cfg == pkg.CFG_LC_STEREO:
        frames = [dict(coeffs=c, ics=i, sbr=None, ps=None) for c, i in synth.lc_stream(rng, 1, 4, 2)]
    else:
        frames = list(synth.he_stream(rng, cfg, 1, 4, hdr))
    out = (C.c_int16 * (192000 // 2))()
    for t, fr in enumerate(frames):
        ics2 = np.zeros(2, pkg.ICS_DT); ics2[: pkg.CORE_CH[cfg]] = fr["ics"][0]
        head = np.zeros(1, np.dtype([("magic", "<u4"), ("cfg", "<u2"), ("flags", "<u2"), ("ics", pkg.ICS_DT, (2,))]))
        head["magic"] = 0x48454141; head["cfg"] = cfg; head["ics"][0] = ics2
        blob = head.tobytes() + fr["coeffs"][0].astype(np.float32).tobytes()
        if fr["sbr"] is not None:
            head["flags"] = 1 if t == 0 else 0
            blob = head.tobytes() + fr["coeffs"][0].astype(np.float32).tobytes() + fr["sbr"][0].tobytes()
            if fr["ps"] is not None:
                blob += fr["ps"][0].tobytes()
            if t == 0:
                blob += hdr[0].tobytes()
        buf = C.create_string_buffer(blob, len(blob))
        pkt = HeaacPacket(C.cast(buf, C.c_void_p), len(blob))
        size = C.c_int(192000)
        used = lib.heaac_codec_decode(C.byref(ctx), out, C.byref(size), C.byref(pkt))
        assert used == len(blob)
        assert size.value == pkg.OUT_LEN[cfg] * pkg.OUT_CH[cfg] * 2
        got = np.frombuffer(out, np.int16, size.value // 2).reshape(pkg.OUT_LEN[cfg], pkg.OUT_CH[cfg])
        if cfg == pkg.CFG_LC_STEREO:
            ref, state = oracle.lc_decode_batch(2, fr["coeffs"], fr["ics"], state, oracle.PCM_S16)
        else:
            ref, state = oracle.he_decode_batch(cfg, fr["coeffs"], fr["ics"], fr["sbr"], hdr, fr["ps"], state,
                                                oracle.PCM_S16)
        assert np.array_equal(got, ref[0]), "frame %d" % t
    # too-small output buffer is refused like avcodec_decode_audio3 does (utils.c:645-651)
    small = C.c_int(1000)
    assert lib.heaac_codec_decode(C.byref(ctx), out, C.byref(small), C.byref(pkt)) == -1
    assert lib.heaac_codec_close(C.byref(ctx)) == 0


def test_codec_refuses_sbr_rates_that_contradict_each_other(pkg):
    """An extension rate strictly between the core rate and twice the core rate: ff_sbr_apply would synthesise 1024
    samples (aacsbr.c:1719) and aac_decode_frame hand out 2048 (aacdec.c:2080-2081).  Refused at open."""
    lib = pkg.lib()
    asc = bytes([0x2B, 0x12, 0x88, 0x00])                  # SBR, 24 kHz core, 32 kHz extension rate, AAC-LC pair
    ctx = HeaacCodecContext(cfg=-1, extradata=asc, extradata_size=len(asc))
    codec = C.c_void_p.in_dll(lib, "heaac_aac_decoder")
    assert lib.heaac_codec_open(C.byref(ctx), C.c_void_p(C.addressof(codec))) == -1


def _adts(au, aot, si, chan):
    """ADTS frame around one raw_data_block (aac_parser.c:29-70)."""
    import aac_bitwriter as W
    bw = W.BitWriter()
    for v, n in ((0xfff, 12), (0, 1), (0, 2), (1, 1), (aot - 1, 2), (si, 4), (0, 1), (chan, 3), (0, 4),
                 (7 + len(au), 13), (0x7ff, 11), (0, 2)):
        bw.put(v, n)
    return bw.bytes(pad=0) + au


@pytest.mark.parametrize("mode", ["lc_asc", "lc_adts_mono", "ltp_profile_adts_mono", "hev1_asc", "hev2_asc", "hev2_implicit", "sbr_too_late",
                                  "hev1_downsampled_asc", "hev2_downsampled_asc"])
def test_codec_decodes_access_units(pkg, oracle, dev, mode):
    """cfg = HEAAC_CFG_FROM_STREAM: the packets are AAC access units, as avcodec_decode_audio3 hands them to the
    reference's aac_decode_frame.  Configuration from extradata or the first ADTS header (whose profile bits may say
    SSR or LTP: parse_adts_frame_header aacdec.c:1935-1971 takes any, and the stream decodes as AAC-LC until an element
    uses the missing tool), SBR explicit or implicit (first access unit only), mono + SBR decoded as Parametric Stereo;
    "downsampled SBR" -- an extension rate equal to the core rate: the 32-band synthesis bank, 1024 samples per frame at
    the core rate (aacsbr.c:1719, aacdec.c:2080-2084).  int16 PCM against the oracle's spectral tools
    + decode on the separately parsed records, state chained over six frames."""
    import test_parse as TP
    import sbr_bitwriter as SW
    lib = pkg.lib()
    rng = np.random.default_rng(hash(mode) % 1000)
    asc = dict(lc_asc=bytes([0x11, 0x90]), lc_adts_mono=None, ltp_profile_adts_mono=None, hev1_asc=bytes([0x2B, 0x11, 0x88, 0x00]),
               hev2_asc=bytes([0xEB, 0x09, 0x88, 0x00]), hev2_implicit=bytes([0x13, 0x08]),
               sbr_too_late=bytes([0x13, 0x08]), hev1_downsampled_asc=bytes([0x2B, 0x13, 0x08, 0x00]),
               hev2_downsampled_asc=bytes([0xEB, 0x0B, 0x08, 0x00]))[mode]
    si = 3 if mode in ("lc_asc", "lc_adts_mono", "ltp_profile_adts_mono") else 6
    aot = 4 if mode == "ltp_profile_adts_mono" else 2
    cpe = mode in ("lc_asc", "hev1_asc", "hev1_downsampled_asc")
    down = "downsampled" in mode
    ch = 2 if cpe else 1
    he = mode.startswith("hev")
    hcfg = {"lc_asc": pkg.CFG_LC_STEREO, "lc_adts_mono": pkg.CFG_LC_MONO, "ltp_profile_adts_mono": pkg.CFG_LC_MONO, "hev1_asc": pkg.CFG_HEV1,
            "hev2_asc": pkg.CFG_HEV2, "hev2_implicit": pkg.CFG_HEV2, "sbr_too_late": pkg.CFG_LC_MONO,
            "hev1_downsampled_asc": pkg.CFG_HEV1, "hev2_downsampled_asc": pkg.CFG_HEV2}[mode]
    length = 1024 if down else pkg.OUT_LEN[hcfg]
    ctx = HeaacCodecContext(cfg=-1, extradata=asc, extradata_size=len(asc) if asc else 0)
    codec = C.c_void_p.in_dll(lib, "heaac_aac_decoder")
    assert lib.heaac_codec_open(C.byref(ctx), C.c_void_p(C.addressof(codec))) == 0
    # the checker's own parse of the same units
    m4 = pkg.AacConfig()
    m4.object_type, m4.sampling_index, m4.sample_rate, m4.chan_config = aot, si, 48000 if si == 3 else 24000, ch
    m4.sbr, m4.ps = (1 if he else 0), (1 if hcfg == pkg.CFG_HEV2 else 0)
    tab = pkg.SbrHeaderTable(64)
    st, sst = np.zeros(1, pkg.AAC_STREAM_DT), pkg.sbr_streams(1)
    writer = SW.SbrStreamWriter(pkg, ch, ps=not cpe)
    state = np.zeros((1, pkg.STATE_WORDS[hcfg]), np.float32)
    ref_rng = np.full(1, 0x1f2e3d4c, np.int32)
    out = (C.c_int16 * (192000 // 2))()
    loud = 0
    for t in range(6):
        sbr_here = he or (mode == "sbr_too_late" and t >= 2)
        if sbr_here:
            import copy
            while True:
                keep = copy.deepcopy((writer.ch, writer.ps, writer.header, writer.hdr_rec, writer.kx_m, writer.coupling))
                bits, exp = writer.frame(rng, new_header=(he and t == 3), respec=(he and t == 3))
                if (4 + len(bits) + 7) // 8 <= 269:                    # one fill element
                    break
                writer.ch, writer.ps, writer.header, writer.hdr_rec, writer.kx_m, writer.coupling = keep
            au, _ = TP._write_au(rng, si, 2, cpe, extras=False, sbr=(bits, False), quiet=True)
        else:
            au, _ = TP._write_au(rng, si, 2, cpe, extras=False, quiet=True)
        pkt_bytes = _adts(au, aot, si, ch) if asc is None else au
        buf = C.create_string_buffer(pkt_bytes, len(pkt_bytes))
        pkt = HeaacPacket(C.cast(buf, C.c_void_p), len(pkt_bytes))
        size = C.c_int(192000)
        used = lib.heaac_codec_decode(C.byref(ctx), out, C.byref(size), C.byref(pkt))
        assert used == len(pkt_bytes), (t, used)                       # only zero padding follows the END element
        assert (ctx.channels, ctx.frame_size) == (pkg.OUT_CH[hcfg], length), t
        # aac_channel_layout[channel_config - 1] (aacdec.c:247): a mono stream says MONO even with two PS channels out
        assert ctx.channel_layout == (0x3 if cpe else 0x4), t
        assert ctx.sample_rate == (24000 if mode == "sbr_too_late" or down else 48000)
        assert size.value == length * pkg.OUT_CH[hcfg] * 2
        got = np.frombuffer(out, np.int16, size.value // 2).reshape(length, pkg.OUT_CH[hcfg]).copy()
        if he:
            p = pkg.heaac_parse_batch(m4, st, sst, tab, [pkt_bytes], threads=1, with_ps=hcfg == pkg.CFG_HEV2)
        else:
            p = pkg.aac_parse_batch(m4, st, [pkt_bytes], threads=1)
        assert p["failed"] == 0
        coeffs = np.ascontiguousarray(p["coeffs"][:, :ch])
        ref_c, ref_rng = oracle.spectral_tools_batch(ch, coeffs, p["tools"], rng=ref_rng)
        ics = np.ascontiguousarray(p["ics"][:, :ch])
        if he:
            ref, state = oracle.he_decode_batch(hcfg, ref_c, ics, p["sbr"], tab.headers(), p["ps"] if hcfg == pkg.CFG_HEV2 else None,
                                                state, oracle.PCM_S16, downsampled=down)
        else:
            ref, state = oracle.lc_decode_batch(ch, ref_c, ics, state, oracle.PCM_S16)
        assert np.array_equal(got, ref[0]), "frame %d" % t
        loud = max(loud, int(np.abs(got.astype(int)).max()))
    assert loud > 50                                                   # audible, not all zeros
    # a packet that is not an access unit of this stream
    junk = bytes([0xff] * 64)
    jb = C.create_string_buffer(junk, len(junk))
    pkt = HeaacPacket(C.cast(jb, C.c_void_p), len(junk))
    size = C.c_int(192000)
    assert lib.heaac_codec_decode(C.byref(ctx), out, C.byref(size), C.byref(pkt)) < 0
    assert lib.heaac_codec_close(C.byref(ctx)) == 0
