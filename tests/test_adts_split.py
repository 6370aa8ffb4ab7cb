"""heaac_adts_probe / heaac_adts_split (csrc/adts_split.c): a raw ADTS buffer, as the reference's aac demuxer and
parser walk it (libavformat/raw.c:666-717, libavcodec/aac_ac3_parser.c:26-100), into access units -- on buffers
made by the test bit writers: clean, behind an ID3v2 tag, behind junk, with a destroyed header, with payload bytes
that look like a sync word, cut short."""
import numpy as np
import pytest

import aac_bitwriter as W
import test_parse as TP


def adts_frame(au, aot=2, si=3, chan=2, crc=False, rdb=0):
    """ADTS header (aac_parser.c:29-70) around one raw_data_block."""
    bw = W.BitWriter()
    hs = 9 if crc else 7
    for v, n in ((0xfff, 12), (0, 1), (0, 2), (0 if crc else 1, 1), (aot - 1, 2), (si, 4), (0, 1), (chan, 3), (0, 4),
                 (hs + len(au), 13), (0x7ff, 11), (rdb, 2)):
        bw.put(v, n)
    return bw.bytes(pad=0) + (b"\x12\x34" if crc else b"") + au


def make_frames(seed, count, si=3, cpe=True):
    rng = np.random.default_rng(seed)
    return [adts_frame(TP._write_au(rng, si, 2, cpe, extras=False, quiet=True)[0], si=si, chan=2 if cpe else 1, crc=(k % 5 == 4))
            for k in range(count)]


def spans(pkg, buf):
    p, h = pkg.adts_split(buf)
    # the packets tile the buffer
    at = 0
    for q in p:
        assert int(q["offset"]) == at
        at += int(q["size"])
    assert at == len(buf)
    return [(int(q["kind"]), int(q["offset"]), int(q["size"])) for q in p], h


def test_clean_stream_is_cut_at_every_header(pkg):
    frames = make_frames(1, 12)
    buf = b"".join(frames)
    got, h = spans(pkg, buf)
    assert [g[0] for g in got] == [pkg.ADTS_FRAME] * 12
    assert [g[2] for g in got] == [len(f) for f in frames]
    assert (h.object_type, h.sampling_index, h.chan_config, h.num_aac_frames) == (2, 3, 2, 1)
    assert pkg.adts_probe(buf) == 51                                    # >= 3 frames in step from the start
    p, _ = pkg.adts_split(buf)
    assert [int(x) for x in p["header_size"]] == [9 if k % 5 == 4 else 7 for k in range(12)]


def test_id3v2_tag_and_junk_prefix(pkg):
    frames = make_frames(2, 6)
    body = bytes(range(1, 100))                                         # no 0xFF in it
    tag = b"ID3\x03\x00\x00" + bytes([0, 0, 0, len(body)]) + body
    buf = tag + b"".join(frames)
    got, _ = spans(pkg, buf)
    assert got[0] == (pkg.ADTS_TAG, 0, len(tag)) and [g[0] for g in got[1:]] == [pkg.ADTS_FRAME] * 6
    assert pkg.adts_probe(buf) == 51
    junk = bytes(np.random.default_rng(3).integers(0, 0xf0, 333, dtype=np.uint8))
    buf = junk + b"".join(frames)
    got, _ = spans(pkg, buf)
    assert got[0] == (pkg.ADTS_JUNK, 0, 333) and [g[0] for g in got[1:]] == [pkg.ADTS_FRAME] * 6
    assert pkg.adts_probe(buf) == 25                                    # a run of >= 3, not from the start


def test_resync_behind_a_destroyed_header(pkg):
    frames = make_frames(4, 9)
    bad = bytearray(frames[4]); bad[0] = 0x00; bad[1] = 0x00              # the sync word is gone
    buf = b"".join(frames[:4]) + bytes(bad) + b"".join(frames[5:])
    got, _ = spans(pkg, buf)
    kinds = [g[0] for g in got]
    assert kinds == [pkg.ADTS_FRAME] * 4 + [pkg.ADTS_JUNK] + [pkg.ADTS_FRAME] * 4
    assert got[4][2] == len(frames[4])                                  # the whole damaged frame is one junk span
    assert [g[2] for g in got[5:]] == [len(f) for f in frames[5:]]


def test_a_sync_word_inside_the_payload_is_not_a_frame(pkg):
    """While searching, a header is only believed if another one follows one frame length on (the probe's rule):
    seven payload bytes that parse as a header do not start a frame."""
    frames = make_frames(5, 8)
    fake = bytes([0xff, 0xf1, 0x4c, 0x80, 0x02, 0x1f, 0xfc])            # sync, LC, 48 kHz, stereo, length 16
    hdr = pkg.adts_parse_header(fake)
    assert hdr[1] == 7 and hdr[0].frame_length == 16
    junk = bytes([1, 2, 3]) + fake + bytes(range(40))
    buf = junk + b"".join(frames)
    got, _ = spans(pkg, buf)
    assert got[0] == (pkg.ADTS_JUNK, 0, len(junk)) and [g[0] for g in got[1:]] == [pkg.ADTS_FRAME] * 8
    # directly behind a good frame the reference's rule applies: the header is taken without looking ahead
    buf = b"".join(frames[:2]) + fake + bytes(9) + b"".join(frames[2:])
    got, _ = spans(pkg, buf)
    assert [g[0] for g in got] == [pkg.ADTS_FRAME] * 9 and got[2][2] == 16


def test_truncated_tail_and_degenerate_buffers(pkg):
    frames = make_frames(6, 4)
    buf = b"".join(frames)[:-10]
    got, _ = spans(pkg, buf)
    assert [g[0] for g in got] == [pkg.ADTS_FRAME] * 3 + [pkg.ADTS_TRUNCATED]
    # while SEARCHING, a header whose frame would run past the end is not believed (it may be payload bytes)
    fake = bytes([0xff, 0xf1, 0x4c, 0x80, 0x7f, 0xff, 0xfc])            # length 1023
    buf = bytes(12) + fake + bytes(100)
    assert [g[0] for g in spans(pkg, buf)[0]] == [pkg.ADTS_JUNK]
    buf = bytes(12) + fake + bytes(1023 - 7)                            # ... unless it ends exactly with the buffer
    assert [g[0] for g in spans(pkg, buf)[0]] == [pkg.ADTS_JUNK, pkg.ADTS_FRAME]
    assert spans(pkg, b"")[0] == []
    assert spans(pkg, bytes(5))[0] == [(pkg.ADTS_JUNK, 0, 5)]
    assert spans(pkg, bytes([0xff] * 64))[0] == [(pkg.ADTS_JUNK, 0, 64)]  # sync words with a reserved sampling index
    assert pkg.adts_probe(bytes(64)) == 0 and pkg.adts_probe(b"") == 0


def test_split_frames_parse_like_the_bare_access_units(pkg):
    """The packets feed the access-unit parser: frame by frame the same records as parsing the raw data blocks."""
    rng = np.random.default_rng(7)
    aus = [TP._write_au(rng, 3, 2, True, extras=False, quiet=True)[0] for _ in range(6)]
    buf = bytes(17) + b"".join(adts_frame(a) for a in aus)
    p, h = pkg.adts_split(buf)
    frames = [buf[int(q["offset"]):int(q["offset"] + q["size"])] for q in p if q["kind"] == pkg.ADTS_FRAME]
    assert len(frames) == 6
    cfg = TP._cfg(pkg)
    assert (cfg.object_type, cfg.sampling_index) == (h.object_type, h.sampling_index)
    st_a, st_b = np.zeros(1, pkg.AAC_STREAM_DT), np.zeros(1, pkg.AAC_STREAM_DT)
    for f, au in zip(frames, aus):
        a = pkg.aac_parse_batch(cfg, st_a, [f], threads=1)
        b = pkg.aac_parse_batch(cfg, st_b, [au], threads=1)
        assert a["failed"] == 0 and b["failed"] == 0
        assert np.array_equal(a["coeffs"].view(np.uint32), b["coeffs"].view(np.uint32))
        assert a["ics"].tobytes() == b["ics"].tobytes() and a["tools"].tobytes() == b["tools"].tobytes()


@pytest.mark.gpu
def test_aac_file_bytes_to_pcm_on_the_gpu(pkg, oracle, dev):
    """A .aac buffer (ID3v2 tag, junk, HE-AACv2 ADTS frames with implicit SBR + PS, one frame with a destroyed
    header) -> heaac_adts_split -> the codec surface, configuration from the first ADTS header as
    parse_adts_frame_header takes it -> int16 PCM; against the oracle fed with the separately parsed records of the
    frames the splitter delivers."""
    import copy
    import ctypes as C
    import sbr_bitwriter as SW
    from test_shim_gpu import HeaacCodecContext, HeaacPacket
    rng = np.random.default_rng(31)
    si = 6                                                              # 24 kHz core, 48 kHz out
    writer = SW.SbrStreamWriter(pkg, 1, ps=True)
    frames = []
    for t in range(9):
        while True:
            keep = copy.deepcopy((writer.ch, writer.ps, writer.header, writer.hdr_rec, writer.kx_m, writer.coupling))
            bits, _ = writer.frame(rng, new_header=(t == 5), respec=(t == 5))
            if (4 + len(bits) + 7) // 8 <= 269:
                break
            writer.ch, writer.ps, writer.header, writer.hdr_rec, writer.kx_m, writer.coupling = keep
        au, _ = TP._write_au(rng, si, 2, False, extras=False, sbr=(bits, False), quiet=True)
        frames.append(adts_frame(au, si=si, chan=1))
    bad = bytearray(frames[3]); bad[1] = 0x0f
    tag = b"ID3\x04\x00\x00\x00\x00\x00\x20" + bytes(32)
    buf = tag + bytes(range(1, 60)) + b"".join(frames[:3]) + bytes(bad) + b"".join(frames[4:])
    p, h = pkg.adts_split(buf)
    assert [int(k) for k in p["kind"]] == [pkg.ADTS_TAG, pkg.ADTS_JUNK] + [pkg.ADTS_FRAME] * 3 + [pkg.ADTS_JUNK] + [pkg.ADTS_FRAME] * 5
    units = [buf[int(q["offset"]):int(q["offset"] + q["size"])] for q in p if q["kind"] == pkg.ADTS_FRAME]

    lib = pkg.lib()
    ctx = HeaacCodecContext(cfg=-1)                                     # no extradata: the stream configures itself
    codec = C.c_void_p.in_dll(lib, "heaac_aac_decoder")
    assert lib.heaac_codec_open(C.byref(ctx), C.c_void_p(C.addressof(codec))) == 0
    m4 = pkg.AacConfig()
    m4.object_type, m4.sampling_index, m4.sample_rate, m4.chan_config, m4.sbr, m4.ps = 2, si, 24000, 1, 1, 1
    tab = pkg.SbrHeaderTable(64)
    st, sst = np.zeros(1, pkg.AAC_STREAM_DT), pkg.sbr_streams(1)
    cfg = pkg.CFG_HEV2
    state = np.zeros((1, pkg.STATE_WORDS[cfg]), np.float32)
    ref_rng = np.full(1, 0x1f2e3d4c, np.int32)
    out = (C.c_int16 * (192000 // 2))()
    loud = refused = 0
    for t, unit in enumerate(units):
        b = C.create_string_buffer(unit, len(unit))
        pkt = HeaacPacket(C.cast(b, C.c_void_p), len(unit))
        size = C.c_int(192000)
        assert lib.heaac_codec_decode(C.byref(ctx), out, C.byref(size), C.byref(pkt)) == len(unit), t
        assert (ctx.channels, ctx.frame_size, ctx.sample_rate) == (2, 2048, 48000)
        got = np.frombuffer(out, np.int16, size.value // 2).reshape(2048, 2).copy()
        q = pkg.heaac_parse_batch(m4, st, sst, tab, [unit], threads=1, with_ps=True)
        # the frame behind the lost one carries deltas against data that never arrived: its SBR payload may be refused
        # (HEAAC_PARSE_ERR_DATA, a `start = 0` record: the core is still decoded, as the reference goes on)
        assert int(q["status"][0]) in (0, -1) and int(q["info"][0]["channels"]) == 1, t
        refused += int(q["status"][0]) != 0
        coeffs = np.ascontiguousarray(q["coeffs"][:, :1])
        ref_c, ref_rng = oracle.spectral_tools_batch(1, coeffs, q["tools"], rng=ref_rng)
        ref, state = oracle.he_decode_batch(cfg, ref_c, np.ascontiguousarray(q["ics"][:, :1]), q["sbr"], tab.headers(),
                                            q["ps"], state, oracle.PCM_S16)
        assert np.array_equal(got, ref[0]), "frame %d" % t
        loud = max(loud, int(np.abs(got.astype(int)).max()))
    assert loud > 50 and refused <= 2
    assert lib.heaac_codec_close(C.byref(ctx)) == 0
