#!/usr/bin/env python3
"""Probe: does the VALU-bound Gaussian overlap with the bandwidth-bound Sobel+NMS when they run on two streams?

128 x 4K resident frames cut into sub-batches; stream A runs the Gaussian of sub-batch i+1 while stream B runs
Sobel+NMS (s16 stage form) of sub-batch i, chained by events.  Compared with the same launches on one stream.
Prints one JSON line.  (Two contexts of the library, each bound to a torch stream.)
"""
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from canny_edge_amd import capi  # noqa: E402
from canny_edge_amd.synth import synth_frame  # noqa: E402


def main():
    H, W, F = 2160, 3840, 128
    dev = torch.device("cuda", 0)
    base = torch.from_numpy(np.stack([synth_frame(H, W, 42 + i) for i in range(16)])).to(dev)
    d_img = base[torch.arange(F, device=dev) % 16].contiguous()
    d_sm = torch.empty((F, H, W), dtype=torch.int16, device=dev)
    d_out = torch.empty((F, H, W), dtype=torch.int16, device=dev)
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    a, b = capi.Context(0), capi.Context(0)
    a.set_stream(sA.cuda_stream)
    b.set_stream(sB.cuda_stream)
    px = H * W
    res = {}

    def run(n_sub, overlap, reps=12):
        per = F // n_sub
        evs = [torch.cuda.Event() for _ in range(n_sub)]

        def once():
            for i in range(n_sub):
                off = i * per * px
                a.dev_gaussian(d_img.data_ptr() + off, 1.4, H, W, per, d_sm.data_ptr() + 2 * off)
                if overlap:
                    evs[i].record(sA)
                    sB.wait_event(evs[i])
                    b.dev_sobel_nms(d_sm.data_ptr() + 2 * off, H, W, per, d_out.data_ptr() + 2 * off)
                else:
                    a.dev_sobel_nms(d_sm.data_ptr() + 2 * off, H, W, per, d_out.data_ptr() + 2 * off)
        for _ in range(3):
            once()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            once()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    t_end = time.perf_counter() + 0.6
    while time.perf_counter() < t_end:
        run(1, False, reps=2)
    for n_sub in (1, 2, 4, 8, 16):
        res[f"sub{n_sub}_one_stream_ms"] = round(run(n_sub, False), 4)
        res[f"sub{n_sub}_two_streams_ms"] = round(run(n_sub, True), 4)
    # sanity: the overlapped result equals the sequential one
    ref = d_out.clone()
    run(4, True, reps=1)
    res["same_result"] = bool(torch.equal(ref, d_out))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
