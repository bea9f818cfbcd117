#!/usr/bin/env python3
"""BASELINE config 3 (1024 x 1080p, sigma 1.0) host->host: pipeline mode x pipelines x chunk size, per-call times."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from canny_edge_amd import capi
from canny_edge_amd.synth import synth_frame

H, W = 1080, 1920
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = capi.Context(0)
base = np.stack([synth_frame(H, W, 100 + i) for i in range(16)])
src = ctx.pinned_array((N, H, W), np.uint8)
for i in range(N):
    src[i] = base[i % 16]
# resident compute time of one default chunk (12 frames)
d_in, d_out = ctx.malloc(12 * H * W), ctx.malloc(12 * H * W * 2)
ctx.h2d(d_in, src[:12])
for _ in range(20):
    ctx.dev_canny(d_in, 1.0, 50, 150, H, W, 12, d_out)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    ctx.dev_canny(d_in, 1.0, 50, 150, H, W, 12, d_out)
ctx.synchronize()
print(f"resident compute of a 12-frame chunk: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms", flush=True)
for u8 in (False, True):
    dst = ctx.pinned_array((N, H, W), np.uint8 if u8 else np.int16)
    for mode, pipes in ((1, 1), (1, 2)):
        for mb in (16, 24, 32, 48, 64):
            ctx.set_option("tune_batch_pipe_mode", mode)
            ctx.set_option("tune_batch_workers", pipes)
            ctx.set_option("tune_batch_chunk_mb", mb)
            ctx.canny_batch(src[:64], 1.0, 50, 150, out=dst[:64], u8=u8)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                ctx.canny_batch(src, 1.0, 50, 150, out=dst, u8=u8)
                ts.append(time.perf_counter() - t0)
            t = min(ts)
            print(f"{N}x1080p {'u8 ' if u8 else 's16'} mode={mode} pipes={pipes} chunk={mb:2d}MB: " +
                  " ".join(f"{x * 1e3:6.1f}" for x in ts) + f" ms  best {N * H * W / t / 1e9:6.2f} Gpix/s  D2H {dst.nbytes / t / 1e9:5.1f} GB/s",
                  flush=True)
    del dst
ctx.close()
