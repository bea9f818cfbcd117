#!/usr/bin/env python3
"""One 4K frame, device resident: canny_hip_dev_canny called N times, each waited for.  Meant to run under
rocprofv3 --kernel-trace --stats to see what each kernel of a single-frame call costs.  Prints the wall time per call."""
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from canny_edge_amd import capi  # noqa: E402
from canny_edge_amd.synth import synth_frame  # noqa: E402

H, W = 2160, 3840
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ctx = capi.Context(0)
img = synth_frame(H, W, 42)
d_in, d_out = ctx.malloc(img.nbytes), ctx.malloc(img.nbytes * 2)
ctx.h2d(d_in, img)
for _ in range(50):
    ctx.dev_canny(d_in, 1.4, 50, 150, H, W, 1, d_out)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    ctx.dev_canny(d_in, 1.4, 50, 150, H, W, 1, d_out)
    ctx.synchronize()
t1 = time.perf_counter()
for _ in range(n):
    ctx.dev_canny(d_in, 1.4, 50, 150, H, W, 1, d_out)
ctx.synchronize()
t2 = time.perf_counter()
print(json.dumps({"waited_ms_per_frame": round((t1 - t0) / n * 1e3, 4), "stream_ms_per_frame": round((t2 - t1) / n * 1e3, 4),
                  "sweeps": ctx.last_hysteresis_iterations}))
