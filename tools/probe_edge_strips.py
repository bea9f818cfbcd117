#!/usr/bin/env python3
"""How much slower are the Sobel+NMS kernel's frame-border strips (COL_EDGE variants) than interior strips?
Times the s16 and the fused kernel on frames of equal area whose share of border strips differs:
  width  480: 1 strip  per row of strips (every wave is a border strip, 60 of 62 owner lanes busy)
  width  992: 2 strips (all border)
  width 3840: 8 strips (2 border)     width 7936: 16 strips (2 border)"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from canny_edge_amd import capi
from canny_edge_amd.synth import synth_frame


def main():
    ctx = capi.Context(0)
    H = 2160
    total_px = 128 * 2160 * 3840
    base = synth_frame(H, 7936, 42)
    for W in (480, 992, 3840, 7936):
        F = total_px // (H * W)
        img = np.ascontiguousarray(base[:, :W])
        d_img = ctx.malloc(F * H * W)
        for i in range(F):
            ctx.h2d(d_img + i * H * W, np.roll(img, 13 * i, axis=0) if i % 16 else img)
        d_sm, d_out = ctx.malloc(F * H * W * 2), ctx.malloc(F * H * W * 2)
        ctx.dev_gaussian(d_img, 1.4, H, W, F, d_sm)
        ctx.synchronize()
        ctx.profile_enable(True)
        res = {"s16": [], "fused": [], "gauss": []}
        for _ in range(7):
            ctx.profile_reset()
            ctx.dev_sobel_nms(d_sm, H, W, F, d_out)
            res["s16"].append(ctx.profile_get(capi.STAGE_SOBEL_NMS)[0])
            ctx.profile_reset()
            ctx.dev_canny(d_img, 1.4, 50, 150, H, W, F, d_out)
            ctx.synchronize()
            res["fused"].append(ctx.profile_get(capi.STAGE_SOBEL_NMS)[0])
            res["gauss"].append(ctx.profile_get(capi.STAGE_GAUSSIAN)[0])
        ctx.profile_enable(False)
        px = F * H * W
        print(f"W={W:5d} F={F:5d}: " + "  ".join(
            f"{k} {statistics.median(v):.4f} ms = {statistics.median(v) * 1e6 / px * 1e3:.4f} ps/px" for k, v in res.items()),
            flush=True)
        for p in (d_img, d_sm, d_out):
            ctx.free(p)
    ctx.close()


if __name__ == "__main__":
    main()
