"""A bit WRITER for AAC raw data blocks (test infrastructure): emits single_channel_element /
channel_pair_element access units from known side info and quantised spectra, following the syntax of
ISO/IEC 14496-3 tables 4.4 - 4.54, so that the product's parser (csrc/aac_parse.c) can be pinned by a
round trip.  It shares no code with the parser: the code tables are read here from the generated header's
TEXT, symbols are emitted with a plain dict lookup."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "ffmpeg-heaac_amd", "csrc", "aac_iso_tables.h")


def _tables():
    txt = open(HDR).read()

    def arr(name):
        m = re.search(r"\b%s\[\d+\]\s*=\s*\{(.*?)\};" % name, txt, re.S)
        return [int(v, 0) for v in re.findall(r"0x[0-9a-fA-F]+|\d+", m.group(1))]
    t = {k: arr(k) for k in ("aac_sf_code", "aac_sf_bits", "aac_spec_first", "aac_spec_code", "aac_spec_bits",
                             "aac_num_swb_1024", "aac_num_swb_128", "aac_swb_first_1024", "aac_swb_offset_1024",
                             "aac_swb_first_128", "aac_swb_offset_128", "aac_tns_max_bands_1024",
                             "aac_tns_max_bands_128", "aac_pred_sfb_max")}
    m = re.search(r"aac_tns_map\[4\]\[16\]\s*=\s*\{(.*?)\n\};", txt, re.S)
    rows = re.findall(r"\{([^{}]*)\}", m.group(1))
    t["tns_map"] = [[float.fromhex(v.strip().rstrip("f")) for v in r.split(",") if v.strip()] for r in rows]
    return t


T = _tables()
MOD = {5: 9, 6: 9, 7: 8, 8: 8, 9: 13, 10: 13, 11: 17}
LAV = {1: 1, 2: 1, 3: 2, 4: 2, 5: 4, 6: 4, 7: 7, 8: 7, 9: 12, 10: 12, 11: 8191}   # largest absolute value per book


class BitWriter:
    def __init__(self):
        self.bits = []

    def put(self, v, n):
        assert 0 <= v < (1 << n) if n else v == 0, (v, n)
        self.bits.extend((v >> (n - 1 - i)) & 1 for i in range(n))

    def align(self):
        while len(self.bits) % 8:
            self.bits.append(0)

    def bytes(self, pad=8):
        b = list(self.bits)
        while len(b) % 8:
            b.append(0)
        out = bytearray(int("".join(map(str, b[i:i + 8])), 2) for i in range(0, len(b), 8))
        return bytes(out) + bytes(pad)            # FF_INPUT_BUFFER_PADDING_SIZE zero bytes (avcodec.h:440)


def swb(si, eight):
    if eight:
        f = T["aac_swb_first_128"][si]; n = T["aac_num_swb_128"][si]
        return T["aac_swb_offset_128"][f:f + n + 1]
    f = T["aac_swb_first_1024"][si]; n = T["aac_num_swb_1024"][si]
    return T["aac_swb_offset_1024"][f:f + n + 1]


def put_spec(bw, book, vals):
    """One codeword of spectral book `book` for a quad / pair of quantised values."""
    first = T["aac_spec_first"][book - 1]
    if book <= 2:
        idx = sum((v + 1) * m for v, m in zip(vals, (27, 9, 3, 1)))
    elif book <= 4:
        idx = sum(abs(v) * m for v, m in zip(vals, (27, 9, 3, 1)))
    elif book <= 6:
        idx = (vals[0] + 4) * 9 + (vals[1] + 4)
    else:
        a = [min(abs(v), 16) if book == 11 else abs(v) for v in vals]
        idx = a[0] * MOD[book] + a[1]
    bw.put(T["aac_spec_code"][first + idx], T["aac_spec_bits"][first + idx])
    if book in (3, 4) or book >= 7:
        for v in vals:
            if v:
                bw.put(1 if v < 0 else 0, 1)
    if book == 11:
        for v in vals:
            a = abs(v)
            if a >= 16:
                n = a.bit_length() - 1 - 4              # escape: N ones, a zero, N + 4 bits of a - 2^(N+4)
                bw.put((1 << n) - 1, n); bw.put(0, 1); bw.put(a - (1 << (n + 4)), n + 4)


def put_sf(bw, delta):
    bw.put(T["aac_sf_code"][delta + 60], T["aac_sf_bits"][delta + 60])


def random_ics(rng, si, aot, allow_intensity, quiet=False):
    """Side info + quantised spectrum of one channel, everything the syntax can carry in this slice.
    quiet: levels of quiet audio (peaks around 1e-3 of full scale) instead of the syntax's extremes, for tests
    that run the SBR stage behind the parser (its energy arithmetic overflows on 1e5 x full scale)."""
    eight = rng.random() < 0.3
    off = swb(si, eight)
    num_swb = len(off) - 1
    d = dict(window_sequence=2 if eight else int(rng.choice([0, 1, 3])), window_shape=int(rng.integers(0, 2)),
             max_sfb=int(rng.integers(1, num_swb + 1)), eight=eight, off=off, num_swb=num_swb)
    if eight:
        d["grouping"] = [int(x) for x in rng.integers(0, 2, 7)]
        lens, cur = [], 1
        for gbit in d["grouping"]:
            if gbit:
                cur += 1
            else:
                lens.append(cur); cur = 1
        lens.append(cur)
        d["group_len"] = lens
    else:
        d["group_len"] = [1]
    d["predictor_present"] = 0
    if not eight and aot == 1 and rng.random() < 0.5:
        d["predictor_present"] = 1
        d["reset_group"] = int(rng.integers(1, 31)) if rng.random() < 0.4 else 0
        d["prediction_used"] = [int(x) for x in rng.integers(0, 2, min(d["max_sfb"], T["aac_pred_sfb_max"][si]))]
    ng, ms = len(d["group_len"]), d["max_sfb"]
    # sections: runs of one band type per group
    bt = np.zeros((ng, ms), int)
    for g in range(ng):
        k = 0
        while k < ms:
            ln = int(rng.integers(1, ms - k + 1))
            choices = [0] + list(range(1, 12)) + [13] + ([14, 15] if allow_intensity else [])
            bt[g, k:k + ln] = int(rng.choice(choices)); k += ln
    d["band_type"] = bt
    d["global_gain"] = int(rng.integers(128, 145)) if quiet else int(rng.integers(100, 180))
    d["sf_delta"] = rng.integers(-2, 3, (ng, ms)) if quiet else rng.integers(-6, 7, (ng, ms))
    # noise gain of the first noise band (sent as 9 bits)
    d["noise_level"] = int(rng.integers(52, 72)) if quiet else int(rng.integers(60, 200))      # amplitude 2^((level - 100) / 4)
    # quantised lines per (group, band): [group_len][width]
    q = {}
    for g in range(ng):
        for i in range(ms):
            w = off[i + 1] - off[i]
            b = bt[g, i]
            if 1 <= b <= 11:
                lav = LAV[b] if b < 11 else (15 if rng.random() < 0.5 else int(rng.choice([40] if quiet else [40, 300, 5000])))
                v = rng.integers(-lav, lav + 1, (d["group_len"][g], w))
                if b == 11 and lav > 15:
                    v[rng.random(v.shape) < 0.7] = 0
                if b in (3, 4, 7, 8, 9, 10, 11) or b in (1, 2, 5, 6):
                    q[(g, i)] = v
    d["q"] = q
    d["pulse"] = None
    if not eight and rng.random() < 0.4:
        npulse = int(rng.integers(1, 5))
        start = int(rng.integers(0, min(num_swb, 20)))
        pos = [off[start] + int(rng.integers(0, 32))]
        for _ in range(npulse - 1):
            pos.append(pos[-1] + int(rng.integers(0, 32)))
        if pos[-1] <= 1023:
            d["pulse"] = dict(swb=start, first=pos[0] - off[start], pos=pos, amp=[int(x) for x in rng.integers(0, 16, npulse)])
    d["tns"] = None
    if rng.random() < 0.5:
        nw = 8 if eight else 1
        t = dict(n_filt=[], coef_res=[], filt=[])
        for w in range(nw):
            nf = int(rng.integers(0, 2 if eight else 4))
            t["n_filt"].append(nf); t["coef_res"].append(int(rng.integers(0, 2)))
            fl = []
            for _ in range(nf):
                order = int(rng.integers(0, 4)) if quiet else int(rng.integers(0, 8 if eight else 13))
                fl.append(dict(length=int(rng.integers(0, num_swb + 1)) if not eight else int(rng.integers(0, 16)),
                               order=order, direction=int(rng.integers(0, 2)), compress=int(rng.integers(0, 2)),
                               idx=None))
                clen = t["coef_res"][-1] + 3 - fl[-1]["compress"]
                fl[-1]["idx"] = [int(x) for x in rng.integers(0, 1 << clen, order)]
                if quiet:                                  # small reflection coefficients: a filter gain of a few
                    fl[-1]["idx"] = [int(x) for x in rng.integers(0, 2, order)]
            t["filt"].append(fl)
        d["tns"] = t
    return d


def put_ics_info(bw, d, si, aot):
    bw.put(0, 1)
    bw.put(d["window_sequence"], 2); bw.put(d["window_shape"], 1)
    if d["eight"]:
        bw.put(d["max_sfb"], 4)
        for gbit in d["grouping"]:
            bw.put(gbit, 1)
    else:
        bw.put(d["max_sfb"], 6)
        bw.put(d["predictor_present"], 1)
        if d["predictor_present"]:
            bw.put(1 if d["reset_group"] else 0, 1)
            if d["reset_group"]:
                bw.put(d["reset_group"], 5)
            for u in d["prediction_used"]:
                bw.put(u, 1)


def put_ics(bw, d, si, aot, common_window):
    """individual_channel_stream(); returns the expected (band_type[128], sf[128]) as the parser must report."""
    bw.put(d["global_gain"], 8)
    if not common_window:
        put_ics_info(bw, d, si, aot)
    ng, ms, eight = len(d["group_len"]), d["max_sfb"], d["eight"]
    nb = 3 if eight else 5
    esc = (1 << nb) - 1
    for g in range(ng):                                    # section_data
        k = 0
        while k < ms:
            b = d["band_type"][g, k]
            e = k
            while e < ms and d["band_type"][g, e] == b:
                e += 1
            bw.put(int(b), 4)
            ln = e - k
            while ln >= esc:
                bw.put(esc, nb); ln -= esc
            bw.put(ln, nb)
            k = e
    # scale_factor_data: three differential chains
    gain, noise, pos = d["global_gain"], d["global_gain"] - 90, 100
    noise_flag = True
    sf_offset = 12 if eight else 0
    exp_sf = np.zeros(128, np.float32)
    idx = 0
    for g in range(ng):
        for i in range(ms):
            b = d["band_type"][g, i]
            dl = int(d["sf_delta"][g, i])
            if b == 0:
                pass
            elif b in (14, 15):
                if not 0 <= pos + dl <= 255:
                    dl = 0
                put_sf(bw, dl); pos += dl
                exp_sf[idx] = np.float32(2.0 ** ((-pos + 300 - 200) / 4.0))
            elif b == 13:
                if noise_flag:
                    noise_flag = False
                    bw.put(d["noise_level"] - noise + 256, 9)      # offset[1] += get_bits(9) - 256
                    noise = d["noise_level"]
                else:
                    if not 0 <= noise + dl <= 255:
                        dl = 0
                    put_sf(bw, dl); noise += dl
                exp_sf[idx] = -np.float32(2.0 ** ((noise + sf_offset + 100 - 200) / 4.0))
            else:
                if not 0 <= gain + dl <= 255:
                    dl = 0
                put_sf(bw, dl); gain += dl
                exp_sf[idx] = -np.float32(2.0 ** ((gain + sf_offset - 200) / 4.0))
            idx += 1
    p = d["pulse"]
    bw.put(1 if p else 0, 1)
    if p:
        bw.put(len(p["pos"]) - 1, 2); bw.put(p["swb"], 6); bw.put(p["first"], 5); bw.put(p["amp"][0], 4)
        for j in range(1, len(p["pos"])):
            bw.put(p["pos"][j] - p["pos"][j - 1], 5); bw.put(p["amp"][j], 4)
    t = d["tns"]
    bw.put(1 if t else 0, 1)
    if t:
        for w in range(8 if eight else 1):
            bw.put(t["n_filt"][w], 1 if eight else 2)
            if t["n_filt"][w]:
                bw.put(t["coef_res"][w], 1)
                for f in t["filt"][w]:
                    bw.put(f["length"], 4 if eight else 6); bw.put(f["order"], 3 if eight else 5)
                    if f["order"]:
                        bw.put(f["direction"], 1); bw.put(f["compress"], 1)
                        clen = t["coef_res"][w] + 3 - f["compress"]
                        for v in f["idx"]:
                            bw.put(v, clen)
    bw.put(0, 1)                                           # gain_control_data_present
    for g in range(ng):                                    # spectral_data
        for i in range(ms):
            b = int(d["band_type"][g, i])
            if not 1 <= b <= 11:
                continue
            v = d["q"][(g, i)]
            step = 4 if b <= 4 else 2
            for w in range(v.shape[0]):
                for k in range(0, v.shape[1], step):
                    put_spec(bw, b, [int(x) for x in v[w, k:k + step]])
    return exp_sf


def mag(q):
    """|q|^(4/3) as the decoder forms it: exact float for q < 16, cbrtf(n) * n beyond."""
    q = abs(int(q))
    if q < 16:
        return np.float32(float(q) ** (4.0 / 3.0))
    return np.float32(np.cbrt(np.float32(q))) * np.float32(q)


def expected_coeffs(d, exp_sf):
    """Dequantised spectrum [1024] (without pulses: see expected_pulse), noise / intensity / zero bands 0."""
    out = np.zeros(1024, np.float32)
    off, ms = d["off"], d["max_sfb"]
    base, idx = 0, 0
    for g, gl in enumerate(d["group_len"]):
        for i in range(ms):
            b = int(d["band_type"][g, i])
            if 1 <= b <= 11:
                v = d["q"][(g, i)]
                s = exp_sf[idx]
                for w in range(gl):
                    for k in range(v.shape[1]):
                        m = mag(v[w, k])
                        neg = v[w, k] < 0
                        if v[w, k] == 0:
                            # The sign of a ZERO line.  The reference multiplies by the scalefactor with its sign bit
                            # flipped by "the sign bit now at the head of the pending ones", and only a non-zero line
                            # consumes one: in the unsigned quad books (VMUL4S, aacdec.c:949-972) a zero line therefore
                            # carries the sign of the next non-zero line of its quad, in book 11 (:1199-1201) the first
                            # line of a pair that of the second; the signed books and VMUL2S (books 7-10) do not.
                            if b in (3, 4):
                                rest = [x for x in v[w, k + 1:(k // 4 + 1) * 4] if x != 0]
                                neg = bool(rest) and rest[0] < 0
                            elif b == 11 and k % 2 == 0:
                                neg = v[w, k + 1] < 0
                        out[base + 128 * w + off[i] + k] = (np.float32(-m) if neg else m) * s
            idx += 1
        base += gl * 128
    return out


def expected_pulse(d, exp_sf, coef):
    """aacdec.c:1222-1236 on the dequantised spectrum."""
    p = d["pulse"]
    if not p:
        return coef
    off = d["off"]
    idx = 0
    f32 = np.float32
    for pos, amp in zip(p["pos"], p["amp"]):
        while off[idx + 1] <= pos:
            idx += 1
        sf = exp_sf[idx] if idx < d["max_sfb"] else np.float32(0)
        if sf != 0 and int(d["band_type"][0, idx]) != 13:
            co = coef[pos]
            ico = f32(-amp)
            if co != 0:
                co = f32(co / sf)
                ico = f32(f32(co / f32(np.sqrt(f32(np.sqrt(f32(abs(co))))))) + (f32(-ico) if co > 0 else ico))
            coef[pos] = f32(f32(np.cbrt(f32(abs(ico))) * ico) * sf)
    return coef
