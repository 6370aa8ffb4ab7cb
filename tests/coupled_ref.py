"""Checker for streams whose program config element names coupling channel elements (test infrastructure): the
layout parser's records (pinned by tests/test_parse_layout.py) through the ORACLE the way aac_decode_frame +
spectral_to_sample order the work (aacdec.c:1999-2075, :1903-1933) -- every element's noise substitution /
prediction / stereo tools where it stands in the access unit (one noise generator), the coupling elements' TNS and
IMDCT before their targets', dependent coupling around a target's TNS, independent coupling behind its IMDCT, in
ascending tag order of the coupling elements, then ff_float_to_int16_interleave_c over the output planes.
Also writes such streams with the test bit writers."""
import numpy as np

import aac_bitwriter as W
import test_parse as TP
import test_parse_layout as TL
import test_parse_wide as TW

SCE, CPE, CCE, LFE = 0, 1, 2, 3


def pce_args(elems, cc_tags):
    front = [(int(t == CPE), g) for t, g in elems if t in (SCE, CPE)]
    return front, [], [], [g for t, g in elems if t == LFE], [(int(k & 1), g) for k, g in enumerate(cc_tags)]


def asc(aot, si, elems, cc_tags, rng, he=False):
    """AudioSpecificConfig with channel configuration 0 and the program config element inside; he: SBR signalled by
    the backward-compatible sync extension behind it (object type 5 in front would leave ps = -1, which the reference
    turns into Parametric Stereo channels for every SCE of the layout, aacdec.c:476-477, :203-206)."""
    bw = W.BitWriter()
    bw.put(aot, 5); bw.put(si, 4); bw.put(0, 4)
    bw.put(0, 3)                                           # GASpecificConfig: 1024 samples, no core coder, no extension
    bw.put(0, 4)                                           # element_instance_tag
    a = pce_args(elems, cc_tags)
    TL.write_pce_body(bw, rng, *a[:4], cc=a[4])
    if he:
        bw.put(0x2b7, 11); bw.put(5, 5); bw.put(1, 1); bw.put(si - 3, 4)      # syncExtensionType, SBR, present, rate
        bw.put(0x548, 11); bw.put(0, 1)                                        # PS signalled absent
        bw.put(0, 16)
    return bw.bytes()


def put_sbr_fill(bw, bits):
    cnt = (4 + len(bits) + 7) // 8
    TL.put_fil_count(bw, cnt)
    bw.put(0xd, 4)
    bw.bits.extend(bits)
    bw.bits.extend([0] * (8 * cnt - 4 - len(bits)))


def write_unit(rng, si, aot, elems, tags, points, quiet=False, payloads=None, cce_payloads=None):
    """One access unit: the output elements in a random order, the coupling elements `tags` anywhere between them;
    payloads: {index into elems: bits of an SBR payload for the fill element behind that element}, cce_payloads:
    {coupling element tag: bits} the same for the coupling elements."""
    order = list(range(len(elems)))
    rng.shuffle(order)
    at = sorted(int(x) for x in rng.integers(0, len(elems) + 1, len(tags)))
    real = [e for e in elems if e[0] != LFE]
    bw = W.BitWriter()
    k = 0
    for pos in range(len(elems) + 1):
        while k < len(tags) and at[k] == pos:
            targets = []
            for _ in range(int(rng.integers(1, 3))):
                t, g = real[int(rng.integers(0, len(real)))]
                targets.append((t, g, int(rng.integers(0, 4)) if t == CPE else 2))
            if rng.random() < 0.3:
                targets.insert(int(rng.integers(0, 2)), (int(rng.integers(0, 2)), 14, 3))       # nobody's
            TW.write_cce(bw, rng, si, aot, tags[k], targets, int(rng.choice(points)), quiet=quiet)
            if cce_payloads and cce_payloads.get(tags[k]) is not None:
                put_sbr_fill(bw, cce_payloads[tags[k]])
            k += 1
        if pos < len(elems):
            TL.write_elem(bw, rng, si, aot, *elems[order[pos]], quiet=quiet)
            bits = payloads.get(order[pos]) if payloads else None
            if bits is not None:
                put_sbr_fill(bw, bits)
    bw.put(7, 3)
    return bw.bytes()


class UnitWriter:
    """Access units of one stream: SBR payloads (HE streams) from per-element writers whose state runs along."""
    def __init__(self, pkg, rng, si, aot, elems, cc_tags, points, he, quiet=None):
        import sbr_bitwriter as SW
        self.rng, self.si, self.aot, self.elems, self.points, self.he = rng, si, aot, elems, points, he
        self.quiet = he if quiet is None else quiet
        self.writers = {k: SW.SbrStreamWriter(pkg, 2 if t == CPE else 1) for k, (t, _) in enumerate(elems) if t != LFE} if he else {}
        self.cce_writers = {g: SW.SbrStreamWriter(pkg, 1) for g in cc_tags} if he and 3 in points else {}

    def _payload(self, w):
        import copy
        while True:
            keep = copy.deepcopy((w.ch, w.ps, w.header, w.hdr_rec, w.kx_m, w.coupling))
            bits, _ = w.frame(self.rng)
            if (4 + len(bits) + 7) // 8 <= 269:            # one fill element
                return bits
            w.ch, w.ps, w.header, w.hdr_rec, w.kx_m, w.coupling = keep

    def unit(self, tags, accept=None, pts=None):
        """accept: a predicate on the bytes (the checker's `parses`), None = take the first."""
        payloads = {k: self._payload(w) for k, w in self.writers.items()}
        cce_payloads = {g: self._payload(w) for g, w in self.cce_writers.items() if g in tags}
        while True:
            au = write_unit(self.rng, self.si, self.aot, self.elems, tags, pts or self.points, quiet=self.quiet,
                            payloads=payloads, cce_payloads=cce_payloads)
            if accept is None or accept(au):
                return au


class Checker:
    def __init__(self, pkg, oracle, m4, layout, aot, he=False):
        self.pkg, self.oracle, self.m4, self.layout, self.aot, self.he = pkg, oracle, m4, layout, aot, he
        self.ne, self.nch = int(layout[0]["n_elements"]), int(layout[0]["channels"])
        self.slot_ch = [int(layout[0]["elem"][e]["channels"]) for e in range(self.ne)]
        self.st = np.zeros(pkg.MAX_ELEMENTS, pkg.AAC_STREAM_DT)
        self.state = [np.zeros((1, pkg.STATE_WORDS[pkg.CFG_HEV1 if c == 2 else pkg.CFG_HEV1_MONO] if he else 512 * c), np.float32)
                      for c in self.slot_ch]
        self.tab, self.sst = pkg.SbrHeaderTable(64), pkg.sbr_streams(self.ne)
        fresh = lambda c: np.tile(np.array([0, 0, 1, 1, 0, 0], np.float32), (1, c * pkg.MAX_PREDICTORS, 1)).reshape(1, -1)
        self.pred = [fresh(c) for c in self.slot_ch]
        self.cpred = [fresh(1) for _ in range(pkg.MAX_CCE)]
        self.cstate = [np.zeros((1, pkg.STATE_WORDS[pkg.CFG_HEV1_MONO] if he else 512), np.float32) for _ in range(pkg.MAX_CCE)]
        self.csst = pkg.sbr_streams(pkg.MAX_CCE)
        self.rng = np.full(1, 0x1f2e3d4c, np.int32)
        self.dependent = self.independent = self.sbr_coupled = 0     # gain lists applied so far; coupling channels through SBR
        self.finite = True                                 # every output plane so far
        self.ltp_skipped = 0                               # dependent gain lists an LTP-profile stream left unapplied

    def parses(self, au):
        return self.pkg.aac_parse_frame_layout(self.m4, self.layout.copy(), self.st.copy(), au, with_cce=True)[0] == 0

    def frame(self, au):
        """Returns (int16 [1024][channels], the parsed records)."""
        pkg, oracle, ne, main = self.pkg, self.oracle, self.ne, self.aot == 1
        r, g = pkg.aac_parse_frame_layout(self.m4, self.layout, self.st, au, with_cce=True)
        assert r == 0, r
        if self.m4.object_type == 4:
            # "Dependent coupling is not supported together with LTP" (apply_dependent_coupling, aacdec.c:1822-1826, returns)
            dep = g["cce"]["coupling_point"] != 3
            self.ltp_skipped += int(g["cce"]["n_links"][dep].sum())
            g["cce"]["n_links"][dep] = 0
        cc = g["cce_coeffs"][None].copy()

        def cce_tools(before):
            for seq in range(pkg.MAX_CCE):
                for k in range(pkg.MAX_CCE):
                    rec = g["cce"][0, k]
                    if rec["present"] and rec["seq"] == seq and rec["outputs_before"] == before:
                        c1, self.rng, pr = oracle.spectral_tools_batch_ex(1, oracle.TOOLS_ALL, cc[:, k:k + 1], g["cce_tools"][k:k + 1],
                                                                          rng=self.rng, pred=self.cpred[k] if main else None)
                        cc[:, k] = c1[:, 0]
                        if main:
                            self.cpred[k] = pr
        pre = [None] * ne
        for seq in range(ne):
            cce_tools(seq)
            e = [i for i in range(ne) if int(g["elem"][i]["seq"]) == seq][0]
            c = self.slot_ch[e]
            pre[e], self.rng, pr = oracle.spectral_tools_batch_ex(c, oracle.TOOLS_PRE, np.ascontiguousarray(g["coeffs"][e:e + 1, :c]),
                                                                  g["tools"][e:e + 1], rng=self.rng, pred=self.pred[e] if main else None)
            if main:
                self.pred[e] = pr
        cce_tools(ne)
        rets = {}
        for k in range(pkg.MAX_CCE):
            if not g["cce"][0, k]["present"]:
                continue
            after = g["cce"][0, k]["coupling_point"] == 3
            if not self.he:
                if after:
                    rets[k], self.cstate[k] = oracle.lc_decode_batch(1, cc[:, k:k + 1], g["cce_ics"][k:k + 1][None], self.cstate[k],
                                                                     oracle.PCM_F32)
                continue
            # the coupling channel's own SBR: read whenever a payload stands behind it, applied when it couples AFTER_IMDCT
            ei = g["cce_elem"][k]
            sbr = None
            if int(ei["sbr_payload_bit"]) >= 0:
                rr, sbr, _, _ = pkg.sbr_parse_payload(self.csst[k], self.tab, self.m4.sample_rate, au, 1, False,
                                                      bit=int(ei["sbr_payload_bit"]), cnt=int(ei["sbr_payload_bytes"]),
                                                      misplaced=bool(ei["sbr_misplaced"]))
                assert rr == 0 and not ei["sbr_misplaced"]
            elif after:
                sbr = pkg.sbr_no_payload(self.csst[k], 1)
            if after:
                rets[k], self.cstate[k] = oracle.he_decode_batch(pkg.CFG_HEV1_MONO, cc[:, k:k + 1], g["cce_ics"][k:k + 1][None], sbr,
                                                                 self.tab.headers(), None, self.cstate[k], oracle.PCM_F32)
                self.sbr_coupled += int(sbr["start"][0])
        planes = [None] * self.nch
        for e in range(ne):
            c = self.slot_ch[e]
            post, _, _ = oracle.spectral_tools_batch_ex(c, oracle.TOOLS_POST, pre[e], g["tools"][e:e + 1], cce=g["cce"][e][None],
                                                        cce_coeffs=cc)
            ics = np.ascontiguousarray(g["ics"][e:e + 1, :c])
            if self.he:
                ei = g["elem"][e]
                if int(ei["sbr_payload_bit"]) >= 0:
                    rr, sbr, _, _ = pkg.sbr_parse_payload(self.sst[e], self.tab, self.m4.sample_rate, au, c, False,
                                                          bit=int(ei["sbr_payload_bit"]), cnt=int(ei["sbr_payload_bytes"]),
                                                          misplaced=bool(ei["sbr_misplaced"]))
                    assert rr == (-1 if ei["sbr_misplaced"] else 0)
                else:
                    sbr = pkg.sbr_no_payload(self.sst[e], c)
                f32, self.state[e] = oracle.he_decode_batch(pkg.CFG_HEV1 if c == 2 else pkg.CFG_HEV1_MONO, post, ics, sbr,
                                                            self.tab.headers(), None, self.state[e], oracle.PCM_F32)
            else:
                f32, self.state[e] = oracle.lc_decode_batch(c, post, ics, self.state[e], oracle.PCM_F32)
            for k in range(pkg.MAX_CCE):
                rec = g["cce"][e, k]
                if not rec["present"]:
                    continue
                if rec["coupling_point"] != 3:
                    self.dependent += int(rec["n_links"])
                    continue
                for l in range(int(rec["n_links"])):
                    # apply_independent_coupling over 1024 << sbr samples (:1849-1862): elementwise, so the plane and the
                    # coupling channel go through the oracle's 1024-sample form piece by piece
                    t, units = int(rec["link"][l]["target_ch"]), f32.shape[2] // 1024
                    cpl = np.zeros(units, pkg.COUPLING_DT)
                    cpl["on"][:, 0] = 1
                    cpl["gain"][:, 0] = rec["link"][l]["gain"][0]
                    plane, _ = oracle.couple_after_imdct_batch(1, f32[0, t].reshape(units, 1, 1024), rets[k].reshape(units, 1024), cpl)
                    f32[0, t] = plane.reshape(-1)
                    self.independent += 1
            for j in range(c):
                planes[int(self.layout[0]["elem"][e]["first_channel"]) + j] = f32[0, j]
        self.finite = self.finite and all(np.isfinite(p).all() for p in planes)
        return oracle.float_to_int16_interleave(planes), g
