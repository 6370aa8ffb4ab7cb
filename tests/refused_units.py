"""Access units the reference refuses half-way, written on purpose, each with what its decoder is left with.

aac_decode_frame leaves its element loop at the first error (aacdec.c:2069-2070) and undoes nothing: by then
decode_ics_info has moved a channel's window history on (or cleared it, where the refusal is its own: the memset at
:650, 687-705), decode_spectrum_and_dequant has drawn the random numbers of the noise bands it passed (:1016-1029)
and apply_prediction has stepped the predictors of a channel that was completed (:1381, :1486-1489).  Every writer
here returns (damaged unit, its undamaged twin, model): the model is derived from what was WRITTEN -- never from the
parser under test --
    history    per channel: None (untouched), or the (window_sequence, window_shape) the decoder now remembers
    draws      numbers drawn from the noise generator
    predicted  the channels whose predictors have been stepped (AAC-Main only; with the twin's records)
"""
import numpy as np

import aac_bitwriter as W
import test_parse as TP

KINDS_SCE = ["reserved_bit", "tns_order", "esc_overflow", "fill_overread", "ours_only"]
KINDS_CPE = ["second_channel_reserved_bit", "common_window_esc_overflow", "ms_present_reserved", "fill_overread"]


def lcg(x, steps):
    """lcg_random (aacdec.c:502-505), `steps` times, on a signed 32-bit state."""
    x = int(x) & 0xffffffff
    for _ in range(steps):
        x = (x * 1664525 + 1013904223) & 0xffffffff
    return x - (1 << 32) if x >= 1 << 31 else x


def noise_lines(d, stop=None):
    """Lines in the noise bands of a written channel, band index < stop (all if None)."""
    total, idx = 0, 0
    for g, gl in enumerate(d["group_len"]):
        for i in range(d["max_sfb"]):
            if (stop is None or idx < stop) and int(d["band_type"][g, i]) == 13:
                total += gl * (d["off"][i + 1] - d["off"][i])
            idx += 1
    return total


def _channel(rng, si, aot, long_only=False, like=None):
    for _ in range(400):
        d = W.random_ics(rng, si, aot, allow_intensity=False, quiet=True) if like is None else TP._redraw_like(rng, like, si, aot, quiet=True)
        if like is not None:
            # (the shared layout's band types may name intensity: not in a channel that stands alone in the model)
            d["band_type"][d["band_type"] >= 14] = 0
        if d["max_sfb"] >= 4 and not (long_only and d["eight"]):
            return d
    raise AssertionError("no channel drawn")


def _with_escape_at_the_end(rng, d):
    """The channel's last band (in the order the spectrum is read) becomes a book-11 band ending in an escape of
    8 ones; some noise bands stand in front of it.  Returns the bit to set, counted from the END of the channel's
    bits, that turns the escape prefix into 9 ones ("ESC overflow", :1187-1190)."""
    ng, ms = len(d["group_len"]), d["max_sfb"]
    bt = d["band_type"]
    bt[bt >= 14] = 0
    bt[ng - 1, ms - 1] = 11
    bt[0, 0] = 13
    if ng > 1:
        bt[ng - 1, 0] = 13
    bt[0, ms // 2] = 13
    q = {}
    for g in range(ng):
        for i in range(ms):
            b = int(bt[g, i])
            if 1 <= b <= 11:
                shape = (d["group_len"][g], d["off"][i + 1] - d["off"][i])
                old = d["q"].get((g, i))
                lav = W.LAV[b] if b < 11 else 15
                q[(g, i)] = old if old is not None and old.shape == shape and np.abs(old).max() <= lav else rng.integers(-lav, lav + 1, shape)
    last = q[(ng - 1, ms - 1)]
    last[-1, -2:] = (8191, 0)                      # escape: 8 ones, a zero, 12 bits -- the channel's last 21 bits
    d["q"] = q
    d["pulse"] = None
    return 13                                       # the zero sits 13 bits from the end


def sce_element(rng, si, aot, kind, tag=0, lfe=False):
    """One single channel (or LFE) element: (bits as written with the damage, bits of the undamaged twin, model).
    kind "good": no damage."""
    bw = W.BitWriter()
    bw.put(3 if lfe else 0, 3); bw.put(tag, 4)
    d = _channel(rng, si, aot, long_only=kind == "tns_order")
    flip = None
    if kind in ("esc_overflow", "ours_only"):
        flip = _with_escape_at_the_end(rng, d)
    good = dict(d)
    if kind == "tns_order":
        d = dict(d)
        d["tns"] = dict(n_filt=[1], coef_res=[0], filt=[[dict(length=3, order=21 if aot == 1 else 13, direction=0, compress=0, idx=[0] * 21)]])
        good["tns"] = None
    W.put_ics(bw, d, si, aot, 0)
    model = dict(history=[(d["window_sequence"], d["window_shape"])], draws=0, predicted=[])
    if kind == "reserved_bit":
        bw.bits[7 + 8] = 1
        model["history"] = [(0, 0)]
    elif kind == "esc_overflow":
        assert bw.bits[-flip] == 0 and all(bw.bits[-flip - 8:-flip])
        bw.bits[-flip] = 1
        model["draws"] = noise_lines(d)
    elif kind in ("fill_overread", "good"):
        model["draws"] = noise_lines(d)
        model["predicted"] = [0] if aot == 1 else []
    elif kind == "ours_only":
        # the unit ends inside its spectrum: this parser stops, the reference's reader runs on into its padding
        model = dict(history=[None], draws=0, predicted=[])
    else:
        assert kind == "tns_order"
    gw = W.BitWriter()
    gw.put(3 if lfe else 0, 3); gw.put(tag, 4)
    W.put_ics(gw, good, si, aot, 0)
    return bw.bits, gw.bits, model


def _bytes(bits, pad=8):
    bw = W.BitWriter()
    bw.bits = list(bits)
    return bw.bytes(pad=pad)


FILL_OVERREAD = [1, 1, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1]     # a fill element of 15 + 255 - 1 bytes that are not there (:2053-2056)
END = [1, 1, 1]


def sce_unit(rng, si, aot, kind):
    bad, good, model = sce_element(rng, si, aot, kind)
    if kind == "fill_overread":
        bad = bad + FILL_OVERREAD
    if kind == "ours_only":
        # the unit ends 8 bits early, inside the 12 bits behind the escape prefix: at most 4 + 7 bits are left to read
        return _bytes(bad[:-8], pad=0), _bytes(good + END), model
    return _bytes(bad), _bytes(good + END), model


def cpe_unit(rng, si, aot, kind):
    bad, good, model = cpe_element(rng, si, aot, kind)
    if kind == "fill_overread":
        bad = bad + FILL_OVERREAD
    return _bytes(bad), _bytes(good + END), model


def cpe_element(rng, si, aot, kind, tag=0):
    bw, gw = W.BitWriter(), W.BitWriter()
    for w in (bw, gw):
        w.put(1, 3); w.put(tag, 4)
    if kind in ("second_channel_reserved_bit", "fill_overread", "good", "ours_only"):
        a, b = _channel(rng, si, aot), _channel(rng, si, aot)
        if kind == "ours_only":
            _with_escape_at_the_end(rng, b)
        for w in (bw, gw):
            w.put(0, 1)
            W.put_ics(w, a, si, aot, 0)
        at = len(bw.bits)
        for w in (bw, gw):
            W.put_ics(w, b, si, aot, 0)
        if kind == "ours_only":
            return bw.bits[:-8], gw.bits, dict(history=[None, None], draws=0, predicted=[])
        if kind in ("fill_overread", "good"):
            model = dict(history=[(a["window_sequence"], a["window_shape"]), (b["window_sequence"], b["window_shape"])],
                         draws=noise_lines(a) + noise_lines(b), predicted=[0, 1] if aot == 1 else [])
        else:
            bw.bits[at + 8] = 1                                 # the second channel's ics_info: reserved bit
            model = dict(history=[(a["window_sequence"], a["window_shape"]), (0, 0)], draws=noise_lines(a),
                         predicted=[0] if aot == 1 else [])
        return bw.bits, gw.bits, model
    a = _channel(rng, si, aot)
    b = _channel(rng, si, aot, like=a)
    a["band_type"][a["band_type"] >= 14] = 0
    flip = _with_escape_at_the_end(rng, b) if kind == "common_window_esc_overflow" else None
    for w in (bw, gw):
        w.put(1, 1)
        W.put_ics_info(w, a, si, aot)
    gw.put(0, 2)
    both = [(a["window_sequence"], a["window_shape"])] * 2       # channel 1 takes channel 0's ics (:1462-1464)
    if kind == "ms_present_reserved":
        bw.put(3, 2)                                            # "ms_present = 3 is reserved" (:1465-1468)
        bw.put(0, 64)
        model = dict(history=both, draws=0, predicted=[])
    else:
        assert kind == "common_window_esc_overflow"
        bw.put(0, 2)
        W.put_ics(bw, a, si, aot, 1)
        W.put_ics(bw, b, si, aot, 1)
        assert bw.bits[-flip] == 0
        bw.bits[-flip] = 1
        # nothing is predicted: in a common-window pair apply_prediction waits for both channels (:1486-1489)
        model = dict(history=both, draws=noise_lines(a) + noise_lines(b), predicted=[])
    W.put_ics(gw, a, si, aot, 1)
    W.put_ics(gw, b, si, aot, 1)
    return bw.bits, gw.bits, model


# ---- a 3.0 layout (channel configuration 3: SCE then CPE) ----
KINDS_3_0 = ["pair_esc_overflow", "pair_second_channel_reserved_bit", "centre_tns_order", "fill_overread", "not_allocated", "ours_only"]


def unit_3_0(rng, si, aot, kind):
    """(damaged unit, undamaged twin, model): the model's history / predicted are per element [SCE, CPE]; an element
    the decoder never reached is None."""
    sce_kind = {"centre_tns_order": "tns_order"}.get(kind, "good")
    cpe_kind = {"pair_esc_overflow": "common_window_esc_overflow", "pair_second_channel_reserved_bit": "second_channel_reserved_bit",
                "ours_only": "ours_only"}.get(kind, "good")
    sb, sg, sm = sce_element(rng, si, aot, sce_kind)
    cb, cg, cm = cpe_element(rng, si, aot, cpe_kind)
    twin = _bytes(sg + cg + END)
    if kind == "centre_tns_order":
        return _bytes(sb + cb + END), twin, dict(elements=[sm, None], draws=0)
    if kind == "ours_only":
        # the pair ends inside its spectrum: nothing of the unit counts, not even the centre element in front
        return _bytes(sb + cb, pad=0), twin, dict(elements=[None, None], draws=0)
    tail = []
    if kind == "fill_overread":
        tail = FILL_OVERREAD
    elif kind == "not_allocated":
        tail = [0, 1, 1, 0, 0, 0, 0]                            # an LFE: channel configuration 3 has none (:2011-2015)
    return _bytes(sb + cb + tail), twin, dict(elements=[sm, cm], draws=sm["draws"] + cm["draws"])
