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
