#!/usr/bin/env python3
"""For divisors S within a few ulps of 1 (the Gaussian's full-window weight), which constants c make the
single instruction fma(a, c, a) equal the IEEE quotient a / S for EVERY float a in [0, 256]?  Exhaustive, on the
GPU (canny_hip_selftest_div_fma)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from canny_edge_amd import capi

f32 = np.float32


def neighbours(x, n=2):
    out = [f32(x)]
    lo = hi = f32(x)
    for _ in range(n):
        lo = np.nextafter(lo, f32(-1))
        hi = np.nextafter(hi, f32(1))
        out += [lo, hi]
    return out


with capi.Context(0) as c:
    one = f32(1)
    divisors = [one]
    d = one
    for _ in range(3):
        d = np.nextafter(d, f32(2))
        divisors.append(d)
    d = one
    for _ in range(4):
        d = np.nextafter(d, f32(0))
        divisors.append(d)
    for S in divisors:
        exact = 1.0 / float(S) - 1.0
        cands = [f32(0)] if S == one else neighbours(f32(exact), 3)
        row = []
        for cc in cands:
            n, worst = c.selftest_div_fma(float(S), float(cc))
            row.append((float(cc), n, worst))
        good = [r for r in row if r[1] == 0 or r[2] < 2.0 ** -100]
        print(f"S={float(S)!r:22} 1/S-1={exact:+.10e}  exact c: {[repr(g[0]) for g in good]}  "
              f"best-bad: {min(row, key=lambda r: r[1])[1:]} ")
