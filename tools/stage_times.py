#!/usr/bin/env python3
"""Per-stage device times of the resident pipeline (HIP events), one JSON line.  Used by tools/ab_stage_times.py.

    python tools/stage_times.py [--frames 128] [--steps 12] [--opt name=value ...]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from canny_edge_amd import capi
from canny_edge_amd.synth import synth_frame


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--sigma", type=float, default=1.4)
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--check", action="store_true",
                    help="print the SHA-256 of frame 0's edge map (an A/B of library builds compares them; parity with "
                         "the oracle is the test-suite's job, tests/)")
    a = ap.parse_args()
    H, W, F = a.height, a.width, a.frames
    ctx = capi.Context(0)
    for o in a.opt:
        k, v = o.split("=")
        ctx.set_option(k, int(v))
    base = np.stack([synth_frame(H, W, 42 + i) for i in range(min(8, F))])
    d_img = ctx.malloc(F * H * W)
    for i in range(F):
        ctx.h2d(d_img + i * H * W, base[i % len(base)])
    d_sm, d_out = ctx.malloc(F * H * W * 2), ctx.malloc(F * H * W * 2)
    out = {"lib": os.path.basename(capi.LIB_PATH), "opts": a.opt, "frames": F}

    def run(fn, n):
        fn()
        ctx.synchronize()
        ctx.profile_reset()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        ctx.synchronize()
        wall = (time.perf_counter() - t0) / n * 1e3
        st = {}
        for sid, name in enumerate(capi.STAGE_NAMES):
            ms, cnt = ctx.profile_get(sid)
            if cnt:
                st[name] = round(ms / n, 4)
        return wall, st

    t_end = time.perf_counter() + 0.5
    while time.perf_counter() < t_end:  # clocks up
        ctx.dev_canny(d_img, a.sigma, 50, 150, H, W, F, d_out)
    ctx.synchronize()
    # un-profiled wall time of the whole pipeline
    w0, _ = run(lambda: ctx.dev_canny(d_img, a.sigma, 50, 150, H, W, F, d_out), a.steps)
    out["canny_wall_ms"] = round(w0, 4)
    ctx.profile_enable(True)
    _, st = run(lambda: ctx.dev_canny(d_img, a.sigma, 50, 150, H, W, F, d_out), a.steps)
    out["canny_stages_ms"] = st
    if a.check:
        import hashlib
        got = np.empty((H, W), np.int16)
        ctx.d2h(got, d_out)
        out["edges_sha256"] = hashlib.sha256(got.tobytes()).hexdigest()[:16]
    ctx.dev_gaussian(d_img, a.sigma, H, W, F, d_sm)
    _, st = run(lambda: ctx.dev_sobel_nms(d_sm, H, W, F, d_out), a.steps)
    out["sobel_nms_s16_ms"] = st.get("sobel_nms")
    try:
        ctx.dev_gaussian_u8(d_img, a.sigma, H, W, F, d_sm)
        _, st = run(lambda: ctx.dev_sobel_nms_u8in(d_sm, H, W, F, d_out), a.steps)
        out["sobel_nms_u8in_ms"] = st.get("sobel_nms")
        _, st = run(lambda: ctx.dev_gaussian_u8(d_img, a.sigma, H, W, F, d_sm), a.steps)
        out["gaussian_u8_ms"] = st.get("gaussian")
    except Exception as e:  # older library variants
        out["u8_error"] = str(e)[:80]
    # one frame at a time: latency of a call that is waited for, and a back-to-back stream of calls
    ctx.profile_enable(False)
    lat = []
    for _ in range(60):
        t0 = time.perf_counter()
        ctx.dev_canny(d_img, a.sigma, 50, 150, H, W, 1, d_out)
        ctx.synchronize()
        lat.append(time.perf_counter() - t0)
    lat.sort()
    out["single_frame_latency_ms"] = round(lat[len(lat) // 2] * 1e3, 4)
    t0 = time.perf_counter()
    for _ in range(200):
        ctx.dev_canny(d_img, a.sigma, 50, 150, H, W, 1, d_out)
    ctx.synchronize()
    out["single_frame_stream_ms"] = round((time.perf_counter() - t0) / 200 * 1e3, 4)
    print(json.dumps(out), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
