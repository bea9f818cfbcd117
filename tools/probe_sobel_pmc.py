#!/usr/bin/env python3
"""Runs canny() (fused Sobel+NMS+classify) and the s16 Sobel+NMS stage 3x per arithmetic variant of the marching
kernel, for rocprofv3 --pmc (the kernel names differ in their last template argument)."""
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
d_sm, d_out = ctx.malloc(F * H * W * 2), ctx.malloc(F * H * W * 2)
ctx.dev_gaussian(d_img, 1.4, H, W, F, d_sm)
for variant in (0, 1):
    ctx.set_option("tune_sobel_variant", variant)
    for _ in range(3):
        ctx.dev_canny(d_img, 1.4, 50, 150, H, W, F, d_out)
        ctx.dev_sobel_nms(d_sm, H, W, F, d_out)
    ctx.synchronize()
ctx.close()
