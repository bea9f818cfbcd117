#!/usr/bin/env python3
"""Single 4K frame, device resident: 30 blocking canny() calls (run it under rocprofv3 --kernel-trace to see the
launch timeline of one call)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canny_edge_amd import capi
from canny_edge_amd.synth import synth_frame

ctx = capi.Context(0)
H, W = 2160, 3840
img = synth_frame(H, W, 42)
d_in, d_out = ctx.malloc(img.nbytes), ctx.malloc(img.nbytes * 2)
ctx.h2d(d_in, img)
for _ in range(5):
    ctx.dev_canny(d_in, 1.4, 50, 150, H, W, 1, d_out)
t0 = time.perf_counter()
for _ in range(30):
    ctx.dev_canny(d_in, 1.4, 50, 150, H, W, 1, d_out)
ctx.synchronize()
print("ms per frame", (time.perf_counter() - t0) / 30 * 1e3)
ctx.close()
