#!/usr/bin/env python3
"""Wall time of the byte-plane Gaussian per window on 64 4K frames (one library; run once per library for an A/B):
    CANNY_HIP_LIB=<lib.so> python tools/probe_gauss_sigma.py [sigma ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from canny_edge_amd import capi
from canny_edge_amd.synth import synth_frame
import oracle
H, W, F = 2160, 3840, 64
sigmas = [float(a) for a in sys.argv[1:]] or [0.5, 1.0, 1.4, 2.0, 2.3, 2.6]
ctx = capi.Context(0)
base = np.stack([synth_frame(H, W, 42 + i) for i in range(4)])
d_img = ctx.malloc(F * H * W)
for i in range(F):
    ctx.h2d(d_img + i * H * W, base[i % 4])
d_sm = ctx.malloc(F * H * W)
for sg in sigmas:
    for _ in range(3):
        ctx.dev_gaussian_u8(d_img, sg, H, W, F, d_sm)
    ctx.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        ctx.dev_gaussian_u8(d_img, sg, H, W, F, d_sm)
    ctx.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / n
    got = np.empty((H, W), np.uint8)
    ctx.d2h(got, d_sm)
    ok = bool(np.array_equal(got.astype(np.int16), oracle.gaussian(base[0], sg)))
    print(f"sigma {sg}: window {len(oracle.gaussian_kernel(sg))}  {ms:.3f} ms per 64 x 4K  parity {ok}", flush=True)
ctx.close()
