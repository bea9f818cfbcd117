#!/usr/bin/env python3
"""Runs the marching Gaussian 3x with the 5-op division and 3x with the fma division (for rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from canny_edge_amd import capi
from canny_edge_amd.synth import synth_frame
H, W, F = 2160, 3840, 64
ctx = capi.Context(0)
base = np.stack([synth_frame(H, W, 42 + i) for i in range(4)])
d_img = ctx.malloc(F * H * W)
for i in range(F):
    ctx.h2d(d_img + i * H * W, base[i % 4])
d_sm = ctx.malloc(F * H * W * 2)
for mode in (0, 1):
    ctx.set_option("gaussian_fma_div", mode)
    for _ in range(3):
        ctx.dev_gaussian(d_img, 1.4, H, W, F, d_sm)
    ctx.synchronize()
ctx.close()
