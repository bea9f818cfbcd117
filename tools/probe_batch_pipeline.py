#!/usr/bin/env python3
"""canny_hip_canny_batch on 1024 x 1080p from / into pinned (or, with the argument `pageable`, ordinary) buffers:
workers (streams) x chunk size sweep."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from canny_edge_amd import capi
from canny_edge_amd.synth import synth_frame

H, W, N = 1080, 1920, 1024
ctx = capi.Context(0)
base = np.stack([synth_frame(H, W, 100 + i) for i in range(16)])
PAGEABLE = len(sys.argv) > 1 and sys.argv[1] == "pageable"
pin_in = np.empty((N, H, W), np.uint8) if PAGEABLE else ctx.pinned_array((N, H, W), np.uint8)
for i in range(N):
    pin_in[i] = base[i % 16]
for u8 in (False, True):
    pin_out = (np.empty if PAGEABLE else ctx.pinned_array)((N, H, W), np.uint8 if u8 else np.int16)
    for workers in (2, 3, 4, 6):
        for mb in (16, 24, 32, 64):
            ctx.set_option("tune_batch_workers", workers)
            ctx.set_option("tune_batch_chunk_mb", mb)
            ctx.canny_batch(pin_in[:64], 1.0, 50, 150, out=pin_out[:64], u8=u8)
            best = 1e9
            for _ in range(2):
                t0 = time.perf_counter()
                ctx.canny_batch(pin_in, 1.0, 50, 150, out=pin_out, u8=u8)
                best = min(best, time.perf_counter() - t0)
            gb = (pin_in.nbytes + pin_out.nbytes) / 1e9
            print(f"u8={int(u8)} workers={workers} chunk={mb:3d} MB: {best * 1e3:7.1f} ms  {N * H * W / best / 1e9:6.2f} Gpix/s  "
                  f"{gb / best:5.1f} GB/s both directions", flush=True)
    del pin_out
ctx.close()
