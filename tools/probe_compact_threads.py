#!/usr/bin/env python3
"""Host->host s16 batch (128 x 4K, pinned input) against the size of the expansion pool of the compact transfer,
and against the plain download (tune_batch_compact = 1).  One line per setting: median of 5 calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
from canny_edge_amd import capi
from canny_edge_amd.synth import synth_frame

H, W, N = 2160, 3840, 128
cpus = capi.device_local_cpus(0)
if cpus:
    s = set()
    for part in cpus.split(","):
        a, _, b = part.partition("-")
        s.update(range(int(a), int(b or a) + 1))
    os.sched_setaffinity(0, s & os.sched_getaffinity(0) or os.sched_getaffinity(0))
ctx = capi.Context(0)
base = np.stack([synth_frame(H, W, 42 + i) for i in range(8)])
src = ctx.pinned_array((N, H, W), np.uint8)
for i in range(N):
    src[i] = base[i % 8]
out_pin = ctx.pinned_array((N, H, W), np.int16)
out_pg = np.zeros((N, H, W), np.int16)


def run(out, label):
    for _ in range(2):
        ctx.canny_batch(src, 1.4, 50, 150, out=out)
    t = []
    for _ in range(5):
        t0 = time.perf_counter()
        ctx.canny_batch(src, 1.4, 50, 150, out=out)
        t.append(time.perf_counter() - t0)
    t.sort()
    print(f"{label:60s} median {t[2] * 1e3:7.2f} ms  {N * H * W / t[2] / 1e9:6.2f} Gpix/s  (min {t[0] * 1e3:.2f} max {t[-1] * 1e3:.2f})", flush=True)


ctx.set_option("tune_batch_compact", 1)
run(out_pin, "plain download, pinned output")
ctx.set_option("tune_batch_compact", 0)
for th in (2, 4, 6, 8, 12, 16, 24, 32):
    ctx.set_option("tune_batch_expand_threads", th)
    run(out_pin, f"compact, {th:2d} expansion threads, pinned output")
ctx.set_option("tune_batch_expand_threads", 12)
run(out_pg, "compact, 12 expansion threads, pageable (ordinary) output")
for mb in (8, 16, 48, 96):
    ctx.set_option("tune_batch_chunk_mb", mb)
    run(out_pin, f"compact, 12 threads, {mb} MB chunks")
ctx.set_option("tune_batch_chunk_mb", 0)
# one frame through the reference's entry point canny() (ordinary numpy arrays, as the utils.h shim passes new[] memory)
one = np.array(base[0])
for compact in (1, 0):
    ctx.set_option("tune_batch_compact", compact)
    ctx.canny(one, 1.4, 50, 150)
    t = []
    for _ in range(20):
        t0 = time.perf_counter()
        ctx.canny(one, 1.4, 50, 150)
        t.append(time.perf_counter() - t0)
    t.sort()
    print(f"canny() on one 4K frame, ordinary host arrays, {'plain copies' if compact else 'batch pipeline of one + compact transfer'}: "
          f"median {t[10] * 1e3:.3f} ms (min {t[0] * 1e3:.3f})", flush=True)
ctx.close()
