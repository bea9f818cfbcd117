#!/usr/bin/env python3
"""Per-call times of canny_batch on 128 x 4K from hipHostMalloc'd and from registered buffers, interleaved, to see
whether the host->host rate depends on the kind of pinned memory, on what ran before, or just varies."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from canny_edge_amd import capi
from canny_edge_amd.synth import synth_frame

cpus = capi.device_local_cpus(0)
if cpus and "--no-bind" not in sys.argv:
    s = set()
    for part in cpus.split(","):
        a, _, b = part.partition("-")
        s.update(range(int(a), int(b or a) + 1))
    os.sched_setaffinity(0, s)
H, W, N = 2160, 3840, 128
ctx = capi.Context(0)
base = np.stack([synth_frame(H, W, 42 + i) for i in range(8)])
pin_in = ctx.pinned_array((N, H, W), np.uint8)
pin_out = ctx.pinned_array((N, H, W), np.int16)
reg_in = np.empty((N, H, W), np.uint8)
reg_out = np.empty((N, H, W), np.int16)
for i in range(N):
    pin_in[i] = base[i % 8]
    reg_in[i] = base[i % 8]
reg_out[...] = 0
ctx.host_register(reg_in)
ctx.host_register(reg_out)
d_in, d_out = ctx.malloc(N * H * W), ctx.malloc(N * H * W * 2)
ctx.h2d(d_in, pin_in)


def burn(seconds):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        ctx.dev_canny(d_in, 1.4, 50, 150, H, W, N, d_out)
    ctx.synchronize()


def run(kind, reps=6):
    src, dst = (pin_in, pin_out) if kind == "hipHostMalloc" else (reg_in, reg_out)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        ctx.canny_batch(src, 1.4, 50, 150, out=dst)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{kind:14s} " + " ".join(f"{t:6.1f}" for t in ts) + "  ms per 128 x 4K batch", flush=True)


for rnd in range(3):
    run("hipHostMalloc")
    run("registered")
print("after 2 s of resident compute:")
burn(2.0)
run("hipHostMalloc")
run("registered")
burn(2.0)
run("registered")
run("hipHostMalloc")
print("u8 maps:")
pin_out8 = ctx.pinned_array((N, H, W), np.uint8)
for _ in range(2):
    ts = []
    for _ in range(6):
        t0 = time.perf_counter()
        ctx.canny_batch(pin_in, 1.4, 50, 150, out=pin_out8, u8=True)
        ts.append((time.perf_counter() - t0) * 1e3)
    print("hipHostMalloc u8 " + " ".join(f"{t:6.1f}" for t in ts), flush=True)
ctx.host_unregister(reg_in)
ctx.host_unregister(reg_out)
ctx.close()
