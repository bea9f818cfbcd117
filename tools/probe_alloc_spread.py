#!/usr/bin/env python3
"""One process: the s16 -> s16 Sobel+NMS pass over 128 4K frames with its two planes (a) in two allocations, (b) in one
slab, (c) in two allocations made after a 3 GB spacer was allocated and freed.  Run it several times: the spread is
between processes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from canny_edge_amd import capi
from canny_edge_amd.synth import synth_frame
import oracle
H, W, F = 2160, 3840, 128
ctx = capi.Context(0)
plane = F * H * W * 2
sm = np.stack([oracle.gaussian(synth_frame(H, W, 42 + i), 1.4) for i in range(2)]).astype(np.int16)
def fill(d):
    for i in range(F):
        ctx.h2d(d + i * H * W * 2, sm[i % 2])
def timed(src, dst, n=10):
    ctx.dev_sobel_nms(src, H, W, F, dst); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): ctx.dev_sobel_nms(src, H, W, F, dst)
    ctx.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n
a, b = ctx.malloc(plane), ctx.malloc(plane)
fill(a)
r1 = timed(a, b); c1 = ctx.probe_copy(a, b, plane, 10)
slab = ctx.malloc(2 * plane)
fill(slab)
r2 = timed(slab, slab + plane); c2 = ctx.probe_copy(slab, slab + plane, plane, 10)
ctx.free(a); ctx.free(b); ctx.free(slab)
sp = ctx.malloc(3 << 30); ctx.free(sp)
a, b = ctx.malloc(plane), ctx.malloc(plane)
fill(a)
r3 = timed(a, b); c3 = ctx.probe_copy(a, b, plane, 10)
print(f"two allocations {r1:.3f} ms (copy {c1:.3f})   one slab {r2:.3f} ms (copy {c2:.3f})   two allocations again {r3:.3f} ms (copy {c3:.3f})   [addresses {a:#x} {b:#x}]", flush=True)
ctx.close()
