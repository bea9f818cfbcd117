#!/usr/bin/env python3
"""canny_hip_canny_batch / _u8 host->host sweep: pipeline mode (1 three streams, 2 one in-order stream per pipeline)
x pipelines x chunk size, pinned and pageable buffers, on 128 x 4K (sigma 1.4) and 256 x 1080p (sigma 1.0).
Prints one line per configuration and the best per (shape, dtype, memory)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from canny_edge_amd import capi
from canny_edge_amd.synth import synth_frame

QUICK = len(sys.argv) > 1 and sys.argv[1] == "quick"
ctx = capi.Context(0)
best = {}
for (H, W, N, sigma) in ((2160, 3840, 128, 1.4), (1080, 1920, 256, 1.0)):
    base = np.stack([synth_frame(H, W, 100 + i) for i in range(8)])
    for mem in ("pinned", "pageable"):
        src = ctx.pinned_array((N, H, W), np.uint8) if mem == "pinned" else np.empty((N, H, W), np.uint8)
        for i in range(N):
            src[i] = base[i % 8]
        for u8 in (False, True):
            dt = np.uint8 if u8 else np.int16
            dst = ctx.pinned_array((N, H, W), dt) if mem == "pinned" else np.empty((N, H, W), dt)
            for mode in (1, 2):
                for workers in ((1, 2, 3, 4) if mem == "pinned" else (2, 4, 6, 8)):
                    for mb in ((8, 24) if QUICK else (8, 16, 24, 48)):
                        ctx.set_option("tune_batch_pipe_mode", mode)
                        ctx.set_option("tune_batch_workers", workers)
                        ctx.set_option("tune_batch_chunk_mb", mb)
                        ctx.canny_batch(src, sigma, 50, 150, out=dst, u8=u8)
                        t = 1e9
                        for _ in range(2):
                            t0 = time.perf_counter()
                            ctx.canny_batch(src, sigma, 50, 150, out=dst, u8=u8)
                            t = min(t, time.perf_counter() - t0)
                        gpix = N * H * W / t / 1e9
                        key = (f"{N}x{W}x{H}", "u8" if u8 else "s16", mem)
                        if key not in best or gpix > best[key][0]:
                            best[key] = (gpix, mode, workers, mb)
                        print(f"{key[0]} {key[1]:3s} {mem:8s} mode={mode} pipes={workers} chunk={mb:2d}MB: {t * 1e3:7.1f} ms "
                              f"{gpix:6.2f} Gpix/s  H2D {src.nbytes / t / 1e9:5.1f} D2H {dst.nbytes / t / 1e9:5.1f} GB/s",
                              flush=True)
            del dst
        del src
print("---- best ----")
for key, (gpix, mode, workers, mb) in best.items():
    print(key, f"{gpix:.2f} Gpix/s  mode={mode} pipes={workers} chunk={mb}MB")
ctx.close()
