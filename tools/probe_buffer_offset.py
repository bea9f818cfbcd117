#!/usr/bin/env python3
"""Does the time of the s16 -> s16 Sobel+NMS pass (and of a plain copy) depend on where its output plane lies relative
to its input plane?  One big allocation; the output starts `delta` bytes behind the end of the input, delta swept.
Repeated `passes` times in shuffled order so that drift does not look like an effect."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from canny_edge_amd import capi
from canny_edge_amd.synth import synth_frame
import oracle
H, W, F = 2160, 3840, 128
ctx = capi.Context(0)
plane = F * H * W * 2
pad = 64 << 20
base = ctx.malloc(2 * plane + pad)
sm = np.stack([oracle.gaussian(synth_frame(H, W, 42 + i), 1.4) for i in range(4)]).astype(np.int16)
for i in range(F):
    ctx.h2d(base + i * H * W * 2, sm[i % 4])
deltas = [0, 256, 1024, 4096, 8192, 16384, 65536, 262144, 1 << 20, (1 << 20) + 4096, 2 << 20, 4 << 20, 6 << 20, 16 << 20, 33 << 20]
res = {d: [] for d in deltas}
cp = {d: [] for d in deltas}
def timed(fn, n=10):
    fn(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    ctx.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n
for p in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    order = deltas[:]; random.Random(p).shuffle(order)
    for d in order:
        out = base + plane + d
        res[d].append(timed(lambda: ctx.dev_sobel_nms(base, H, W, F, out)))
        cp[d].append(ctx.probe_copy(base, out, plane, 10))
for d in deltas:
    print(f"delta {d:>9}: sobel+nms s16 {min(res[d]):.3f} .. {max(res[d]):.3f} ms   copy {min(cp[d]):.3f} .. {max(cp[d]):.3f} ms", flush=True)
ctx.close()
