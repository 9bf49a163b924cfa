"""Shared helpers for the tests (no product code)."""
from __future__ import annotations

import numpy as np


class MT19937:
    """std::mt19937 (32-bit Mersenne Twister, init_genrand seeding) -- needed to regenerate the
    input sequence of KAT-0, which is defined in terms of it."""

    def __init__(self, seed: int):
        self.mt = [0] * 624
        self.mt[0] = seed & 0xFFFFFFFF
        for i in range(1, 624):
            self.mt[i] = (1812433253 * (self.mt[i - 1] ^ (self.mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self.idx = 624

    def __call__(self) -> int:
        if self.idx >= 624:
            mt = self.mt
            for i in range(624):
                y = (mt[i] & 0x80000000) | (mt[(i + 1) % 624] & 0x7FFFFFFF)
                mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
            self.idx = 0
        y = self.mt[self.idx]
        self.idx += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF


def kat0_coo():
    """KAT-0 input: rows=1000, cols=900, 20 000 entries, every ~10th in row 7."""
    g = MT19937(1)
    rows, cols = 1000, 900
    r, c, v = [], [], []
    for _ in range(20000):
        rr = 7 if g() % 10 == 0 else g() % rows
        r.append(rr)
        c.append(g() % cols)
        v.append(np.float32(g() % 1000) / np.float32(1000.0) + np.float32(0.001))
    return rows, cols, np.array(r, np.int32), np.array(c, np.int32), np.array(v, np.float32)


def bwd_err(y, y64, mag):
    """max_i |y_i - y64_i| / (|alpha| sum_j |a_ij x_j| + |beta b_i|): the backward-error form of the
    1e-5 gate (BASELINE.json north_star; SURVEY.md section 7, hard part 1)."""
    mag = np.maximum(mag, np.finfo(np.float64).tiny)
    return float(np.max(np.abs(np.asarray(y, np.float64) - y64) / mag))
