#!/usr/bin/env python3
"""Three launches of the byte-plane Gaussian per row-pass form (tune_gaussian_variant 3 = product-fetching, 0 = systolic
with rotated look-ups) over 128 4K frames, for rocprofv3 --pmc / --kernel-trace (tools/pmc_gauss.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from canny_edge_amd import capi
from canny_edge_amd.synth import synth_frame
H, W, F = 2160, 3840, 128
ctx = capi.Context(0)
base = np.stack([synth_frame(H, W, 42 + i) for i in range(4)])
d_img = ctx.malloc(F * H * W)
for i in range(F):
    ctx.h2d(d_img + i * H * W, base[i % 4])
d_sm = ctx.malloc(F * H * W)
for variant in (3, 0):
    ctx.set_option("tune_gaussian_variant", variant)
    for _ in range(3):
        ctx.dev_gaussian_u8(d_img, 1.4, H, W, F, d_sm)
    ctx.synchronize()
ctx.close()
