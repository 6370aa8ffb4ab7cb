"""Channel layouts with several output elements per access unit (csrc/aac_parse.c: heaac_aac_layout_default,
heaac_aac_layout_from_pce, heaac_asc_layout, heaac_aac_parse_frame_layout) against what the reference does
(aacdec.c:113-183 get_che, :192-276 che_configure / output_configure, :303-357 decode_pce, :359-398
set_default_channel_config; aacdectab.h:72-93): the expectations below are stated from those tables and rules, the
access units come from the test bit writer."""
import numpy as np
import pytest

import aac_bitwriter as W
import test_parse as TP

SCE, CPE, CCE, LFE = 0, 1, 2, 3
# channel configuration -> (elements in output order, channel mask, elements in bitstream order)
CONFIGS = {
    1: ([(SCE, 0)], 0x4, [(SCE, 0)]),
    2: ([(CPE, 0)], 0x3, [(CPE, 0)]),
    3: ([(CPE, 0), (SCE, 0)], 0x7, [(SCE, 0), (CPE, 0)]),
    4: ([(CPE, 0), (SCE, 0), (SCE, 1)], 0x107, [(SCE, 0), (CPE, 0), (SCE, 1)]),
    5: ([(CPE, 0), (SCE, 0), (CPE, 1)], 0x37, [(SCE, 0), (CPE, 0), (CPE, 1)]),
    6: ([(CPE, 0), (SCE, 0), (LFE, 0), (CPE, 1)], 0x3f, [(SCE, 0), (CPE, 0), (CPE, 1), (LFE, 0)]),
    7: ([(CPE, 0), (SCE, 0), (LFE, 0), (CPE, 2), (CPE, 1)], 0xff, [(SCE, 0), (CPE, 0), (CPE, 1), (CPE, 2), (LFE, 0)]),
}


def slots(l):
    return [(int(e["type"]), int(e["id"])) for e in l[0]["elem"][:int(l[0]["n_elements"])]]


def test_default_layouts_have_the_reference_order(pkg):
    for cc, (out, mask, _) in CONFIGS.items():
        r, l = pkg.aac_layout_default(cc)
        assert r == 0 and slots(l) == out and int(l[0]["channel_layout"]) == mask
        first = 0
        for e, (t, _) in zip(l[0]["elem"], out):
            assert int(e["first_channel"]) == first and int(e["channels"]) == (2 if t == CPE else 1)
            first += 2 if t == CPE else 1
        assert int(l[0]["channels"]) == first == (8 if cc == 7 else cc)
    for bad in (0, 8, -1, 15):
        assert pkg.aac_layout_default(bad)[0] == -1


def write_elem(bw, rng, si, aot, typ, tag, quiet=True):
    """One SCE / CPE / LFE; returns (channel dicts, expected scalefactors, common_window info)."""
    ch = [W.random_ics(rng, si, aot, allow_intensity=False, quiet=quiet)]
    bw.put(typ, 3); bw.put(tag, 4)
    if typ != CPE:
        return ch, [W.put_ics(bw, ch[0], si, aot, 0)], None
    common = int(rng.integers(0, 2))
    bw.put(common, 1)
    cw = dict(common=common, ms_present=0, ms_mask=np.zeros(128, np.uint8))
    if common:
        ch.append(TP._redraw_like(rng, ch[0], si, aot, quiet))
        W.put_ics_info(bw, ch[0], si, aot)
        cw["ms_present"] = int(rng.integers(0, 3))
        bw.put(cw["ms_present"], 2)
        nb = len(ch[0]["group_len"]) * ch[0]["max_sfb"]
        if cw["ms_present"] == 1:
            cw["ms_mask"][:nb] = rng.integers(0, 2, nb)
            for v in cw["ms_mask"][:nb]:
                bw.put(int(v), 1)
        elif cw["ms_present"] == 2:
            cw["ms_mask"][:nb] = 1
    else:
        ch.append(W.random_ics(rng, si, aot, allow_intensity=True, quiet=quiet))
    return ch, [W.put_ics(bw, ch[0], si, aot, common), W.put_ics(bw, ch[1], si, aot, common)], cw


def write_fill(bw, rng, ext, cnt):
    """fill_element with `cnt` payload bytes of extension type `ext`; returns the bit position behind the type."""
    bw.put(6, 3)
    if cnt >= 15:
        bw.put(15, 4); bw.put(cnt - 14, 8)
    else:
        bw.put(cnt, 4)
    if not cnt:
        return -1
    bw.put(ext, 4)
    at = len(bw.bits)
    for _ in range(8 * cnt - 4):
        bw.put(int(rng.integers(0, 2)), 1)
    return at


def write_dse(bw, rng):
    bw.put(4, 3); bw.put(int(rng.integers(0, 16)), 4)
    align = int(rng.integers(0, 2)); cnt = int(rng.integers(0, 5))
    bw.put(align, 1); bw.put(cnt, 8)
    if align:
        bw.align()
    for _ in range(cnt):
        bw.put(int(rng.integers(0, 256)), 8)


def build(rng, si, aot, elems, sbr_prob=0.0, extras=True, payloads=None, lead=None):
    """elems: [(type, tag)] in bitstream order; payloads: per element the bits of a real SBR payload (or None);
    lead: a function writing something in front (a program config element).  Returns (bytes, [per element dict])."""
    bw = W.BitWriter()
    out = []
    if lead is not None:
        lead(bw)
    if extras and rng.random() < 0.3:
        write_dse(bw, rng)
    for k, (typ, tag) in enumerate(elems):
        ch, sf, cw = write_elem(bw, rng, si, aot, typ, tag)
        e = dict(type=typ, tag=tag, ch=ch, sf=sf, cw=cw, sbr_bit=-1, sbr_bytes=0, sbr_crc=0)
        if payloads is not None and payloads[k] is not None:
            bits = payloads[k]
            cnt = (4 + len(bits) + 7) // 8
            bw.put(6, 3)
            if cnt >= 15:
                bw.put(15, 4); bw.put(cnt - 14, 8)
            else:
                bw.put(cnt, 4)
            bw.put(0xd, 4)
            e.update(sbr_bit=len(bw.bits), sbr_bytes=cnt)
            bw.bits.extend(bits)
            bw.bits.extend([0] * (8 * cnt - 4 - len(bits)))
        elif rng.random() < sbr_prob:
            crc = int(rng.integers(0, 2))
            cnt = int(rng.integers(1, 24))
            e.update(sbr_bit=write_fill(bw, rng, 0xe if crc else 0xd, cnt), sbr_bytes=cnt, sbr_crc=crc)
            if rng.random() < 0.3:
                write_fill(bw, rng, 0x1, int(rng.integers(0, 6)))       # plain fill behind the SBR payload
        elif extras and rng.random() < 0.2:
            write_dse(bw, rng)
        out.append(e)
    bw.put(7, 3)
    return bw.bytes(), out


def check_slot(got, slot, e, si, seq):
    rec = got["elem"][slot]
    assert (int(rec["present"]), int(rec["type"]), int(rec["tag"]), int(rec["seq"])) == (1, e["type"], e["tag"], seq)
    assert int(rec["sbr_payload_bit"]) == e["sbr_bit"]
    if e["sbr_bit"] >= 0:
        assert int(rec["sbr_payload_bytes"]) == e["sbr_bytes"] and int(rec["sbr_crc"]) == e["sbr_crc"]
        # the SBR reader is handed the type of the element in front (aacdec.c:2059) and refuses an LFE's (aacsbr.c:996-1000)
        assert int(rec["sbr_misplaced"]) == int(e["type"] == LFE or e.get("sbr_misplaced", 0))
    wrapped = dict(tools=got["tools"][slot:slot + 1], ics=got["ics"][slot:slot + 1], coeffs=got["coeffs"][slot:slot + 1])
    for c, d in enumerate(e["ch"]):
        TP._check_channel(wrapped, 0, c, d, e["sf"][c], si)
    if e["cw"] is not None:
        t = got["tools"][slot]
        assert int(t["common_window"]) == e["cw"]["common"] and int(t["ms_present"]) == e["cw"]["ms_present"]
        nb = len(e["ch"][0]["group_len"]) * e["ch"][0]["max_sfb"]
        assert np.array_equal(t["ms_mask"][:nb], e["cw"]["ms_mask"][:nb])


@pytest.mark.parametrize("cc", [3, 4, 5, 6, 7])
def test_elements_are_taken_by_position_and_keep_their_tags(pkg, cc):
    rng = np.random.default_rng(900 + cc)
    si, aot = 3, 2
    cfg = TP._cfg(pkg, aot, si, cc)
    out_order, _, arrive = CONFIGS[cc]
    for trial in range(12):
        r, l = pkg.aac_layout_default(cc)
        st = np.zeros(pkg.MAX_ELEMENTS, pkg.AAC_STREAM_DT)
        # tags of the stream: anything, distinct per type
        tags = {}
        for t in (SCE, CPE, LFE):
            n = sum(1 for a in arrive if a[0] == t)
            tags[t] = [int(x) for x in rng.choice(16, n, replace=False)]
        used = {SCE: 0, CPE: 0, LFE: 0}
        elems = []
        for t, _ in arrive:
            elems.append((t, tags[t][used[t]])); used[t] += 1
        want_slot = [out_order.index(a) for a in arrive]
        prev = None
        for frame in range(4):
            order = list(range(len(elems)))
            if frame >= 2:
                rng.shuffle(order)                                 # once mapped, the tag decides, not the position
            au, exp = build(rng, si, aot, [elems[i] for i in order], sbr_prob=0.5)
            r, got = pkg.aac_parse_frame_layout(cfg, l, st, au)
            assert r == 0, (cc, trial, frame, r)
            assert int(got["info"][0]["channels"]) == (8 if cc == 7 else cc)
            for seq, (i, e) in enumerate(zip(order, exp)):
                check_slot(got, want_slot[i], e, si, seq)
                if prev is not None:
                    for c, d in enumerate(e["ch"]):
                        # a common window hands channel 1 channel 0's whole ics_info, its previous window sequence
                        # included; only the previous window SHAPE stays channel 1's own (decode_cpe, :1462-1464)
                        src = 0 if (c == 1 and e["cw"]["common"]) else c
                        assert int(got["ics"][want_slot[i], c]["window_sequence"][1]) == prev[i]["ch"][src]["window_sequence"]
                        assert int(got["ics"][want_slot[i], c]["use_kb_window"][1]) == prev[i]["ch"][c]["window_shape"]
            prev = {i: e for i, e in zip(order, exp)}
            assert int(l[0]["tags_mapped"]) == len(elems)


def test_duplicate_tags_move_up_and_an_sce_may_stand_for_the_lfe(pkg):
    rng = np.random.default_rng(31)
    si, aot = 3, 2
    # "Some buggy encoders appear to set all elem_ids to zero" (:115-127): the second pair with tag 0 becomes tag 1
    r, l = pkg.aac_layout_default(5)
    st = np.zeros(pkg.MAX_ELEMENTS, pkg.AAC_STREAM_DT)
    au, exp = build(rng, si, aot, [(SCE, 0), (CPE, 0), (CPE, 0)], extras=False)
    r, got = pkg.aac_parse_frame_layout(TP._cfg(pkg, aot, si, 5), l, st, au)
    assert r == 0
    exp[2]["tag"] = 1
    for seq, (slot, e) in enumerate(zip([1, 0, 2], exp)):
        check_slot(got, slot, e, si, seq)
    # 5.1 coded as SCE CPE CPE SCE (:146-152): the last SCE lands on the LFE's place, and may carry SBR as an SCE
    r, l = pkg.aac_layout_default(6)
    au, exp = build(rng, si, aot, [(SCE, 0), (CPE, 0), (CPE, 1), (SCE, 1)], sbr_prob=1.0, extras=False)
    r, got = pkg.aac_parse_frame_layout(TP._cfg(pkg, aot, si, 6), l, st, au)
    assert r == 0
    for seq, (slot, e) in enumerate(zip([1, 0, 3, 2], exp)):
        check_slot(got, slot, e, si, seq)
    assert int(got["elem"][2]["sbr_payload_bit"]) >= 0
    # a real LFE with a payload behind it: the reference's SBR reader is handed the LFE's type and switches SBR off
    r, l = pkg.aac_layout_default(6)
    au, exp = build(rng, si, aot, [(SCE, 0), (CPE, 0), (CPE, 1), (LFE, 0)], sbr_prob=1.0, extras=False)
    r, got = pkg.aac_parse_frame_layout(TP._cfg(pkg, aot, si, 6), l, st, au)
    assert r == 0 and int(got["elem"][2]["sbr_payload_bit"]) >= 0 and int(got["elem"][2]["sbr_misplaced"]) == 1
    assert int(got["elem"][3]["sbr_payload_bit"]) >= 0 and int(got["elem"][3]["sbr_misplaced"]) == 0


def test_what_a_layout_has_no_place_for_is_refused(pkg):
    rng = np.random.default_rng(32)
    si, aot = 3, 2
    st = np.zeros(pkg.MAX_ELEMENTS, pkg.AAC_STREAM_DT)

    def parse(cc, elems, **kw):
        r, l = pkg.aac_layout_default(cc)
        au, _ = build(rng, si, aot, elems, extras=False, **kw)
        return pkg.aac_parse_frame_layout(TP._cfg(pkg, aot, si, cc), l, st.copy(), au)[0]
    assert parse(3, [(CPE, 0), (SCE, 0)]) == -1            # the pair where the centre belongs
    assert parse(3, [(SCE, 0), (CPE, 0), (CPE, 1)]) == -1  # a third element
    assert parse(5, [(SCE, 0), (CPE, 0), (SCE, 1)]) == -1  # 4.0's back centre in a 5.0 stream
    assert parse(6, [(SCE, 0), (CPE, 0), (LFE, 0)]) == -1  # the LFE before the back pair
    assert parse(2, [(SCE, 0)]) == -1
    assert parse(5, [(SCE, 0), (CPE, 0)]) == 0             # an access unit may leave elements out
    # a coupling element in a channel-configuration stream: get_che has no place for it (:132-177), whoever asks
    bw = W.BitWriter()
    write_elem(bw, rng, si, aot, SCE, 0)
    import test_parse_wide as TW
    TW.write_cce(bw, rng, si, aot, 0, [(0, 0, 2)], 0)
    bw.put(7, 3)
    for with_cce in (False, True):
        r, l = pkg.aac_layout_default(3)
        assert pkg.aac_parse_frame_layout(TP._cfg(pkg, aot, si, 3), l, st.copy(), bw.bytes(), with_cce=with_cce)[0] == -1
    # an SBR payload in front of every element, and one behind a data stream element
    bw = W.BitWriter()
    write_fill(bw, rng, 0xd, 5); write_elem(bw, rng, si, aot, SCE, 0); bw.put(7, 3)
    r, l = pkg.aac_layout_default(3)
    assert pkg.aac_parse_frame_layout(TP._cfg(pkg, aot, si, 3), l, st.copy(), bw.bytes())[0] == -1
    bw = W.BitWriter()
    write_elem(bw, rng, si, aot, SCE, 0); write_dse(bw, rng); at = write_fill(bw, rng, 0xd, 5); bw.put(7, 3)
    r, l = pkg.aac_layout_default(3)
    r, got = pkg.aac_parse_frame_layout(TP._cfg(pkg, aot, si, 3), l, st.copy(), bw.bytes())
    assert r == 0 and int(got["elem"][1]["sbr_payload_bit"]) == at and int(got["elem"][1]["sbr_misplaced"]) == 1
    # a second payload for the same element (the reference would run its SBR reader twice over one frame's state)
    bw = W.BitWriter()
    write_elem(bw, rng, si, aot, SCE, 0); write_fill(bw, rng, 0xd, 5); write_fill(bw, rng, 0xe, 7); bw.put(7, 3)
    r, l = pkg.aac_layout_default(3)
    assert pkg.aac_parse_frame_layout(TP._cfg(pkg, aot, si, 3), l, st.copy(), bw.bytes())[0] == -3
    # a failed unit leaves the window history alone
    r, l = pkg.aac_layout_default(3)
    s0 = st.copy(); s0["window_sequence"][:] = 2; s0["use_kb_window"][:] = 1
    au, _ = build(rng, si, aot, [(SCE, 0), (CPE, 0)], extras=False)
    s1 = s0.copy()
    assert pkg.aac_parse_frame_layout(TP._cfg(pkg, aot, si, 3), l, s1, au[:len(au) // 2])[0] < 0
    assert s1.tobytes() == s0.tobytes()


def write_drc(bw, rng):
    """dynamic_range_info() behind its type nibble (decode_dynamic_range, aacdec.c:1596-1641); returns its length in
    bytes, the nibble's byte included."""
    bw.put(0xb, 4)
    n, bands = 1, 1
    f = int(rng.integers(0, 2)); bw.put(f, 1)
    if f:
        bw.put(int(rng.integers(0, 16)), 4); bw.put(0, 4); n += 1
    f = int(rng.integers(0, 2)); bw.put(f, 1)
    if f:
        rounds = int(rng.integers(1, 4))
        for k in range(rounds):
            bw.put(int(rng.integers(0, 128)), 7); bw.put(int(k < rounds - 1), 1)
        n += rounds
    f = int(rng.integers(0, 2)); bw.put(f, 1)
    if f:
        incr = int(rng.integers(0, 5))
        bw.put(incr, 4); bw.put(int(rng.integers(0, 16)), 4); n += 1
        bands += incr
        for _ in range(bands):
            bw.put(int(rng.integers(0, 256)), 8)
        n += bands
    f = int(rng.integers(0, 2)); bw.put(f, 1)
    if f:
        bw.put(int(rng.integers(0, 128)), 7); bw.put(0, 1); n += 1
    for _ in range(bands):
        bw.put(int(rng.integers(0, 256)), 8)
    return n + bands


def put_fil_count(bw, cnt):
    bw.put(6, 3)
    if cnt >= 15:
        bw.put(15, 4); bw.put(cnt - 14, 8)
    else:
        bw.put(cnt, 4)


@pytest.mark.parametrize("layout", [False, True])
def test_fill_elements_hold_several_payloads_and_sbr_goes_to_the_element_in_front(pkg, layout):
    """aac_decode_frame :2050-2060 walks a fill element payload by payload; dynamic range control says its own length
    (decode_dynamic_range :1596-1641), every other payload takes what is left.  An SBR payload goes to the channel
    element last seen, its reader is handed the type of the element DIRECTLY in front (:2059) and switches SBR off for
    anything but an SCE / CPE (aacsbr.c:996-1000): reported as `sbr_misplaced`."""
    rng = np.random.default_rng(40 + layout)
    si, aot = 3, 2

    def parse(au):
        if layout:
            r, l = pkg.aac_layout_default(3)
            r, got = pkg.aac_parse_frame_layout(TP._cfg(pkg, aot, si, 3), l, np.zeros(pkg.MAX_ELEMENTS, pkg.AAC_STREAM_DT), au)
            e = got["elem"][1]                             # the SCE of a 3.0 layout
            return r, int(e["sbr_payload_bit"]), int(e["sbr_payload_bytes"]), int(e["sbr_crc"]), int(e["sbr_misplaced"]), got
        r, got = pkg.aac_parse_frame_ex(TP._cfg(pkg, aot, si, 1), np.zeros(1, pkg.AAC_STREAM_DT), au, coeff_channels=1)
        i = got["info"][0]
        return r, int(i["sbr_payload_bit"]), int(i["sbr_payload_bytes"]), int(i["sbr_crc"]), int(i["sbr_misplaced"]), got

    def noise(bw, nbits):
        for _ in range(nbits):
            bw.put(int(rng.integers(0, 2)), 1)
    for trial in range(60):
        case = trial % 6
        bw = W.BitWriter()
        write_elem(bw, rng, si, aot, SCE, 0)
        want = None
        if case == 0:       # dynamic range control alone, exactly as long as it says; the SBR payload in the next fill element
            body = W.BitWriter(); n = write_drc(body, rng)
            put_fil_count(bw, n); bw.bits.extend(body.bits)
            at = write_fill(bw, rng, 0xd, 9)
            want = (at, 9, 0, 1)                           # ... whose predecessor is a fill element: SBR off
        elif case == 1:     # dynamic range control, then the SBR payload inside the SAME fill element
            body = W.BitWriter(); n = write_drc(body, rng)
            extra = int(rng.integers(2, 12))
            put_fil_count(bw, n + extra); bw.bits.extend(body.bits)
            crc = int(rng.integers(0, 2))
            bw.put(0xe if crc else 0xd, 4)
            at = len(bw.bits)
            noise(bw, 8 * extra - 4)
            want = (at, extra, crc, 0)                     # directly behind the SCE
        elif case == 2:     # two dynamic range payloads and plain fill in one element; no SBR
            body = W.BitWriter(); n = write_drc(body, rng) + write_drc(body, rng)
            extra = int(rng.integers(1, 6))
            put_fil_count(bw, n + extra); bw.bits.extend(body.bits)
            bw.put(int(rng.choice([0, 1, 2, 5])), 4); noise(bw, 8 * extra - 4)
        elif case == 3:     # dynamic range control longer than its fill element says: the reader ends up past it
            body = W.BitWriter(); n = write_drc(body, rng)
            short = int(rng.integers(1, n + 1)) if n > 1 else 1
            put_fil_count(bw, short); bw.bits.extend(body.bits)
            if short == n:
                at = write_fill(bw, rng, 0xd, 4); want = (at, 4, 0, 1)
            else:
                crc_at = write_fill(bw, rng, 0xe, 6); want = (crc_at, 6, 1, 1)
        elif case == 4:     # a data stream element between the element and its payload
            write_dse(bw, rng)
            at = write_fill(bw, rng, 0xd, 7); want = (at, 7, 0, 1)
        else:               # the ordinary case
            at = write_fill(bw, rng, 0xd, 11); want = (at, 11, 0, 0)
            write_fill(bw, rng, 0x0, int(rng.integers(0, 5)))
        if layout:
            write_elem(bw, rng, si, aot, CPE, 0)
        bw.put(7, 3)
        r, bit, nbytes, crc, misplaced, got = parse(bw.bytes())
        assert r == 0, (trial, r)
        assert (bit, nbytes, crc, misplaced) == (want if want else (-1, 0, 0, 0)), (trial, case)
        assert int(got["info"][0]["bits_consumed"]) == len(bw.bits)
    # a second SBR payload for the same element; one in front of every channel element
    bw = W.BitWriter()
    write_elem(bw, rng, si, aot, SCE, 0); write_fill(bw, rng, 0xd, 5); write_fill(bw, rng, 0xd, 5)
    if layout:
        write_elem(bw, rng, si, aot, CPE, 0)
    bw.put(7, 3)
    assert parse(bw.bytes())[0] == -3
    bw = W.BitWriter()
    write_fill(bw, rng, 0xd, 5); write_elem(bw, rng, si, aot, SCE, 0)
    if layout:
        write_elem(bw, rng, si, aot, CPE, 0)
    bw.put(7, 3)
    assert parse(bw.bytes())[0] == -1


def write_pce_body(bw, rng, front, side, back, lfe, cc=()):
    """program_config_element behind its instance tag; front / side / back: [(is_cpe, tag)], lfe: [tag]."""
    bw.put(int(rng.integers(0, 4)), 2); bw.put(3, 4)
    for g in (front, side, back):
        bw.put(len(g), 4)
    assoc = int(rng.integers(0, 3))
    bw.put(len(lfe), 2); bw.put(assoc, 3); bw.put(len(cc), 4)
    for _ in range(2):
        f = int(rng.integers(0, 2)); bw.put(f, 1)
        if f:
            bw.put(int(rng.integers(0, 16)), 4)
    f = int(rng.integers(0, 2)); bw.put(f, 1)
    if f:
        bw.put(int(rng.integers(0, 8)), 3)
    for g in (front, side, back):
        for is_cpe, tag in g:
            bw.put(is_cpe, 1); bw.put(tag, 4)
    for tag in lfe:
        bw.put(tag, 4)
    for _ in range(assoc):
        bw.put(int(rng.integers(0, 16)), 4)
    for ind, tag in cc:
        bw.put(ind, 1); bw.put(tag, 4)
    bw.align()
    n = int(rng.integers(0, 6))
    bw.put(n, 8)
    for _ in range(n):
        bw.put(int(rng.integers(0, 256)), 8)


def test_a_program_config_element_gives_the_layout(pkg):
    rng = np.random.default_rng(33)
    si, aot = 3, 2
    for trial in range(40):
        # distinct tags per type
        sce = [int(x) for x in rng.choice(16, int(rng.integers(0, 4)), replace=False)]
        cpe = [int(x) for x in rng.choice(16, int(rng.integers(0, 4)), replace=False)]
        lfe = [int(x) for x in rng.choice(16, int(rng.integers(0, 3)), replace=False)]
        if not sce and not cpe:
            sce = [2]
        members = [(0, t) for t in sce] + [(1, t) for t in cpe]
        rng.shuffle(members)
        cut = sorted(int(x) for x in rng.integers(0, len(members) + 1, 2))
        front, side, back = members[:cut[0]], members[cut[0]:cut[1]], members[cut[1]:]
        lead = int(rng.integers(0, 9))
        bw = W.BitWriter()
        for _ in range(lead):
            bw.put(int(rng.integers(0, 2)), 1)
        write_pce_body(bw, rng, front, side, back, lfe, cc=[(1, 3)] * int(rng.integers(0, 2)))
        end = len(bw.bits)
        bw.put(0x5a, 8)
        r, l, used = pkg.aac_layout_from_pce(bw.bytes(), lead)
        assert r == 0 and used == end - lead
        # output_configure without a channel configuration (:253-268): ids ascending; per id SCE, CPE, LFE
        want = []
        for i in range(16):
            for t, have in ((SCE, sce), (CPE, cpe), (LFE, lfe)):
                if i in have:
                    want.append((t, i))
        assert slots(l) == want and int(l[0]["channel_layout"]) == 0 and int(l[0]["chan_config"]) == 0
        assert int(l[0]["channels"]) == len(sce) + 2 * len(cpe) + len(lfe)
        # elements are found by (type, tag), in any order; one the element does not name is refused
        st = np.zeros(pkg.MAX_ELEMENTS, pkg.AAC_STREAM_DT)
        order = list(range(len(want)))
        rng.shuffle(order)
        au, exp = build(rng, si, aot, [want[i] for i in order], sbr_prob=0.3)
        r, got = pkg.aac_parse_frame_layout(TP._cfg(pkg, aot, si, 0), l, st, au)
        assert r == 0
        for seq, (i, e) in enumerate(zip(order, exp)):
            check_slot(got, i, e, si, seq)
        missing = [(t, i) for t in (SCE, CPE, LFE) for i in range(16) if (t, i) not in want]
        au, _ = build(rng, si, aot, [want[0], missing[int(rng.integers(0, len(missing)))]], extras=False)
        assert pkg.aac_parse_frame_layout(TP._cfg(pkg, aot, si, 0), l, st, au)[0] == -1
    # truncated
    assert pkg.aac_layout_from_pce(bw.bytes()[:3], lead)[0] == -2


def test_coupling_elements_of_a_program_config_layout(pkg):
    """decode_pce names the coupling elements (:343-344), decode_cce reads them (:1503-1570), apply_channel_coupling
    (:1870-1898) finds the gain lists of every output element by (type, place in che[type][] = tag): the record of a
    coupling element comes back once per output slot with the lists that land there."""
    import test_parse_wide as TW
    rng = np.random.default_rng(36)
    si, aot = 3, 2
    elems = [(SCE, 4), (CPE, 1), (CPE, 6), (LFE, 2)]
    cc_tags = [3, 9, 12]                                   # the third is beyond HEAAC_MAX_CCE
    bw = W.BitWriter()
    write_pce_body(bw, rng, [(0, 4), (1, 1)], [], [(1, 6)], [2], cc=[(1, 9), (0, 12), (0, 3)])
    r, l0, _ = pkg.aac_layout_from_pce(bw.bytes(), 0)
    assert r == 0 and slots(l0) == [(CPE, 1), (LFE, 2), (SCE, 4), (CPE, 6)]
    assert [int(l0[0]["tag_map"][CCE][t]) for t in cc_tags] == [1, 2, 3] and int(l0[0]["tag_map"][CCE].astype(bool).sum()) == 3
    want = slots(l0)
    cfg = TP._cfg(pkg, aot, si, 0)
    st = np.zeros(pkg.MAX_ELEMENTS, pkg.AAC_STREAM_DT)
    prev = {}
    points_seen, linked = set(), 0
    for frame in range(30):
        l = l0.copy()
        present = [t for t in cc_tags[:2] if rng.random() < 0.8]
        rng.shuffle(present)
        order = list(range(len(elems)))
        rng.shuffle(order)
        # where the coupling elements stand among the output elements
        at = sorted(int(x) for x in rng.integers(0, len(elems) + 1, len(present)))
        bw = W.BitWriter()
        exp_e, exp_c = [], {}
        k = 0
        for pos in range(len(elems) + 1):
            while k < len(present) and at[k] == pos:
                targets = []
                for _ in range(int(rng.integers(1, 4))):
                    t, tag = elems[int(rng.integers(0, 3))] if rng.random() < 0.8 else (int(rng.integers(0, 2)), 15)
                    targets.append((t, tag, int(rng.integers(0, 4)) if t == CPE else 2))
                point = int(rng.choice([0, 1, 3]))
                exp_c[present[k]] = TW.write_cce(bw, rng, si, aot, present[k], targets, point) + (targets, point, pos, k)
                k += 1
            if pos < len(elems):
                typ, tag = elems[order[pos]]
                ch, sf, cw = write_elem(bw, rng, si, aot, typ, tag)
                exp_e.append(dict(type=typ, tag=tag, ch=ch, sf=sf, cw=cw, sbr_bit=-1, sbr_bytes=0, sbr_crc=0))
        bw.put(7, 3)
        au = bw.bytes()
        r, got = pkg.aac_parse_frame_layout(cfg, l, st, au, with_cce=True)
        if r == -3:         # more than MAX_CCE_LINKS lists of one element on one target
            assert any(len(TW.expected_links(c[4], t == CPE, tag, c[2])) > pkg.MAX_CCE_LINKS for c in exp_c.values() for t, tag in want)
            continue
        assert r == 0 and int(got["info"][0]["n_cce"]) == len(present)
        for seq, (i, e) in enumerate(zip(order, exp_e)):
            check_slot(got, want.index(elems[i]), e, si, seq)
        for k, tag in enumerate(cc_tags[:2]):
            if tag not in exp_c:
                assert not got["cce"][:, k]["present"].any()
                continue
            d, exp_sf, lists, num_gain, targets, point, pos, seq = exp_c[tag]
            points_seen.add(point)
            cw = dict(tools=got["cce_tools"][k:k + 1], ics=got["cce_ics"][k:k + 1][None].repeat(2, 1),
                      coeffs=np.stack([got["cce_coeffs"][k], np.zeros(1024, np.float32)])[None])
            TP._check_channel(cw, 0, 0, d, exp_sf, si)
            if tag in prev:
                assert int(got["cce_ics"][k]["window_sequence"][1]) == prev[tag]["window_sequence"]
                assert int(got["cce_ics"][k]["use_kb_window"][1]) == prev[tag]["window_shape"]
            prev[tag] = d
            for slot, (t, etag) in enumerate(want):
                rec = got["cce"][slot, k]
                eseq = order.index(elems.index((t, etag)))
                assert (rec["present"], rec["elem_id"], rec["coupling_point"], rec["seq"], rec["outputs_before"]) == (1, tag, point, seq, pos)
                assert int(rec["behind_target"]) == int(pos > eseq)
                links = TW.expected_links(targets, t == CPE, etag, lists) if t != LFE else []
                assert int(rec["n_links"]) == len(links)
                linked += len(links)
                for n, (tch, gl) in enumerate(links):
                    assert int(rec["link"][n]["target_ch"]) == tch
                    assert np.array_equal(rec["link"][n]["gain"].view(np.uint32), gl.view(np.uint32))
    assert points_seen == {0, 1, 3} and linked > 20
    # without the coupling records the element is outside the entry; the third coupling element of the layout is taken
    # like the first two (HEAAC_MAX_CCE = 16 since round 4: every tag the syntax has); one the program did not name is
    # not allocated
    def one(tag, with_cce):
        bw = W.BitWriter()
        write_elem(bw, rng, si, aot, SCE, 4)
        TW.write_cce(bw, rng, si, aot, tag, [(0, 4, 2)], 0)
        bw.put(7, 3)
        return pkg.aac_parse_frame_layout(cfg, l0.copy(), st.copy(), bw.bytes(), with_cce=with_cce)[0]
    assert one(3, True) == 0 and one(3, False) == -3 and one(12, True) == 0 and one(5, True) == -1 and one(5, False) == -1
    # an SBR payload behind a coupling element is that element's own (decode_extension_payload hands it to che_prev,
    # :2059; read_sbr_data takes TYPE_CCE as a single channel, aacsbr.c:986): reported in cce_elem, or outside the entry
    bw = W.BitWriter()
    write_elem(bw, rng, si, aot, SCE, 4)
    TW.write_cce(bw, rng, si, aot, 9, [(0, 4, 2)], 3)
    at = write_fill(bw, rng, 0xe, 9)
    write_dse(bw, rng)
    TW.write_cce(bw, rng, si, aot, 3, [(0, 4, 2)], 0)
    write_fill(bw, rng, 0x0, 2)
    at2 = write_fill(bw, rng, 0xd, 5)                      # a fill element in between: the reader will switch this one's SBR off
    bw.put(7, 3)
    r, got = pkg.aac_parse_frame_layout(cfg, l0.copy(), st.copy(), bw.bytes(), with_cce=True)
    assert r == 0 and int(got["elem"][2]["sbr_payload_bit"]) == -1
    e9, e3 = got["cce_elem"][1], got["cce_elem"][0]
    assert (int(e9["present"]), int(e9["type"]), int(e9["tag"]), int(e9["seq"])) == (1, CCE, 9, 0)
    assert (int(e9["sbr_payload_bit"]), int(e9["sbr_payload_bytes"]), int(e9["sbr_crc"]), int(e9["sbr_misplaced"])) == (at, 9, 1, 0)
    assert (int(e3["tag"]), int(e3["seq"]), int(e3["sbr_payload_bit"]), int(e3["sbr_misplaced"])) == (3, 1, at2, 1)
    assert pkg.aac_parse_frame_layout(cfg, l0.copy(), st.copy(), bw.bytes(), with_cce="no_sbr")[0] == -3
    # the same coupling element twice: the second moves up to the next tag (get_che :121-127), which nobody named
    bw = W.BitWriter()
    write_elem(bw, rng, si, aot, SCE, 4)
    TW.write_cce(bw, rng, si, aot, 3, [(0, 4, 2)], 0); TW.write_cce(bw, rng, si, aot, 3, [(0, 4, 2)], 0)
    bw.put(7, 3)
    assert pkg.aac_parse_frame_layout(cfg, l0.copy(), st.copy(), bw.bytes(), with_cce=True)[0] == -1


def test_a_self_configuring_stream_finds_its_program_config_element(pkg):
    """aac_decode_frame evaluates a program config element wherever it stands ahead of the channel elements while
    nothing is configured (:2036-2046); data stream and fill elements in front of it are read past."""
    from test_shim_gpu import _adts
    rng = np.random.default_rng(37)
    si, aot = 3, 2
    for trial in range(20):
        bw = W.BitWriter()
        for _ in range(int(rng.integers(0, 3))):
            write_dse(bw, rng) if rng.random() < 0.5 else write_fill(bw, rng, int(rng.choice([0, 1, 2])), int(rng.integers(0, 20)))
        if rng.random() < 0.5:
            body = W.BitWriter(); n = write_drc(body, rng)
            put_fil_count(bw, n); bw.bits.extend(body.bits)
        bw.put(5, 3); bw.put(int(rng.integers(0, 16)), 4)
        write_pce_body(bw, rng, [(0, 0), (1, 0)], [], [(1, 1)], [0], cc=[(1, 7)])
        write_elem(bw, rng, si, aot, SCE, 0)
        bw.put(7, 3)
        au = bw.bytes()
        for pkt in (au, _adts(au, aot, si, 0)):
            r, l = pkg.aac_layout_from_au(pkt)
            assert r == 0 and slots(l) == [(SCE, 0), (CPE, 0), (LFE, 0), (CPE, 1)] and int(l[0]["tag_map"][CCE][7]) == 1
    # a channel element first (nothing is allocated), an SBR payload first, no program at all, a truncated unit
    bw = W.BitWriter(); write_elem(bw, rng, si, aot, SCE, 0); bw.put(7, 3)
    assert pkg.aac_layout_from_au(bw.bytes())[0] == -1
    bw = W.BitWriter(); write_fill(bw, rng, 0xd, 6); bw.put(5, 3); bw.put(0, 4); write_pce_body(bw, rng, [(0, 0)], [], [], []); bw.put(7, 3)
    assert pkg.aac_layout_from_au(bw.bytes())[0] == -1
    bw = W.BitWriter(); write_dse(bw, rng); bw.put(7, 3)
    assert pkg.aac_layout_from_au(bw.bytes())[0] == -1
    assert pkg.aac_layout_from_au(au[:2])[0] in (-1, -2)


def test_audio_specific_config_with_and_without_a_program_config_element(pkg):
    rng = np.random.default_rng(34)
    for cc in range(1, 8):
        bw = W.BitWriter()
        bw.put(2, 5); bw.put(3, 4); bw.put(cc, 4); bw.put(0, 3)
        r, c, l = pkg.asc_layout(bw.bytes())
        assert r == 0 and c.chan_config == cc and slots(l) == CONFIGS[cc][0]
    bw = W.BitWriter()
    bw.put(2, 5); bw.put(4, 4); bw.put(0, 4)              # AAC-LC, 44.1 kHz, channel configuration 0
    bw.put(0, 1); bw.put(1, 1); bw.put(0x155, 14); bw.put(0, 1)      # 1024 samples, core coder delay, no extension
    bw.put(9, 4)                                           # element_instance_tag
    write_pce_body(bw, rng, [(0, 0), (1, 0)], [], [(1, 1)], [0])
    r, c, l = pkg.asc_layout(bw.bytes())
    assert r == 0 and c.chan_config == 0 and c.sampling_index == 4
    assert slots(l) == [(SCE, 0), (CPE, 0), (LFE, 0), (CPE, 1)] and int(l[0]["channels"]) == 6     # C L R LFE Ls Rs (:246-250)
    # 960-sample frames are refused as at aac_decode_init
    bw = W.BitWriter()
    bw.put(2, 5); bw.put(3, 4); bw.put(6, 4); bw.put(1, 1); bw.put(0, 2)
    assert pkg.asc_layout(bw.bytes())[0] == -3


def test_codec_refuses_the_layout_the_reference_would_decode_with_ps_channels(pkg):
    """Explicit SBR with channel configuration 0 leaves ps = -1 in the configuration (mpeg4audio.c:137-139: no channel
    count to rule it out), which the reference turns into ps = 1 (aacdec.c:476-477) and then gives every SCE of the
    program-config layout a second output channel (:203-206).  The codec surface refuses such a stream at open, before it
    touches a device."""
    import ctypes as C
    from test_shim_gpu import HeaacCodecContext
    rng = np.random.default_rng(35)
    bw = W.BitWriter()
    bw.put(5, 5); bw.put(6, 4); bw.put(0, 4); bw.put(3, 4); bw.put(2, 5); bw.put(0, 3); bw.put(0, 4)
    write_pce_body(bw, rng, [(0, 0), (1, 0)], [], [(1, 1)], [0])
    asc = bw.bytes()
    r, c, l = pkg.asc_layout(asc)
    assert r == 0 and c.sbr == 1 and c.ps == -1 and int(l[0]["channels"]) == 6
    lib = pkg.lib()
    ctx = HeaacCodecContext(cfg=-1, extradata=asc, extradata_size=len(asc))
    codec = C.c_void_p.in_dll(lib, "heaac_aac_decoder")
    assert lib.heaac_codec_open(C.byref(ctx), C.c_void_p(C.addressof(codec))) == -1
    assert not ctx.priv_data and not ctx.codec
