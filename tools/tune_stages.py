#!/usr/bin/env python3
"""A/B timing of individual stages in ONE process with interleaved rounds (cdna guide rule 24).

    python tools/tune_stages.py [--frames 64] [--rounds 5]

Prints the median / min device time (HIP events on the launch stream) of the fused Sobel+NMS kernel for
each tuning configuration, plus the other stages of the pipeline for reference."""
import argparse
import statistics
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

from canny_edge_amd import capi
from canny_edge_amd.synth import synth_frame


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--sigma", type=float, default=1.4)
    args = ap.parse_args()
    H, W, F = args.height, args.width, args.frames
    ctx = capi.Context(0)
    base = np.stack([synth_frame(H, W, 42 + i) for i in range(min(8, F))])
    d_img = ctx.malloc(F * H * W)
    for i in range(F):
        ctx.h2d(d_img + i * H * W, base[i % len(base)])
    d_sm = ctx.malloc(F * H * W * 2)
    d_out = ctx.malloc(F * H * W * 2)
    ctx.dev_gaussian(d_img, args.sigma, H, W, F, d_sm)
    ctx.synchronize()
    ctx.profile_enable(True)

    def time_stage(fn, stage):
        ctx.profile_reset()
        fn()
        ms, n = ctx.profile_get(stage)
        return ms / max(n, 1)

    # (name, unused dynamic LDS in KB per workgroup [64 KB -> 2 waves/SIMD instead of 3], rows per segment)
    # (name, unused, rows per segment; 0 = the launcher's own choice)
    configs = [("segauto", 0, 0), ("seg32", 0, 32), ("seg48", 0, 48), ("seg64", 0, 64), ("seg80", 0, 80),
               ("seg96", 0, 96), ("seg128", 0, 128), ("seg180", 0, 180), ("seg256", 0, 256), ("seg360", 0, 360)]
    res = {name: [] for name, _, _ in configs}
    for _ in range(args.rounds):
        for name, _unused, seg in configs:
            ctx.set_option("tune_sobel_seg", seg)
            res[name].append(time_stage(lambda: ctx.dev_sobel_nms(d_sm, H, W, F, d_out), capi.STAGE_SOBEL_NMS))
    alg = 4.0 * F * H * W
    print(f"fused Sobel+NMS, {F} x {W}x{H}, algorithmic {alg / 1e9:.3f} GB per launch")
    for name, _, _ in configs:
        v = res[name]
        med, mn = statistics.median(v), min(v)
        print(f"  {name:14s} median {med:.4f} ms  min {mn:.4f} ms   {alg / med / 1e6:.0f} GB/s  ({alg / med / 8e9 * 100:.1f}% of 8 TB/s)")
    ctx.set_option("tune_sobel_seg", 0)

    # 8 vs 4 pixels per lane: s16 kernel alone and the fused kernel inside canny()
    pres = {}
    for _ in range(args.rounds):
        for px in (0, 1):
            ctx.set_option("tune_sobel_px", px)
            pres.setdefault(("s16", px), []).append(
                time_stage(lambda: ctx.dev_sobel_nms(d_sm, H, W, F, d_out), capi.STAGE_SOBEL_NMS))
            ctx.profile_reset()
            ctx.dev_canny(d_img, args.sigma, 50, 150, H, W, F, d_out)
            ctx.synchronize()
            pres.setdefault(("fused", px), []).append(ctx.profile_get(capi.STAGE_SOBEL_NMS)[0])
    ctx.set_option("tune_sobel_px", 0)
    for (kind, px), v in sorted(pres.items()):
        print(f"Sobel+NMS {kind:5s} {'4' if px else '8'} px/lane: median {statistics.median(v):.4f} ms  min {min(v):.4f} ms")

    # fused kernel: plane bytes staged in LDS (default) vs stored directly
    sres = {0: [], 1: []}
    for _ in range(args.rounds):
        for direct in (0, 1):
            ctx.set_option("tune_plane_stores", direct)
            ctx.profile_reset()
            ctx.dev_canny(d_img, args.sigma, 50, 150, H, W, F, d_out)
            ctx.synchronize()
            sres[direct].append(ctx.profile_get(capi.STAGE_SOBEL_NMS)[0])
    ctx.set_option("tune_plane_stores", 0)
    for direct in (0, 1):
        print(f"Sobel+NMS fused, plane bytes {'stored directly' if direct else 'staged in LDS  '}: median "
              f"{statistics.median(sres[direct]):.4f} ms  min {min(sres[direct]):.4f} ms")

    gmodes = [(variant, fma) for variant in (0, 2, 1) for fma in (0, 1)]
    gres = {m: [] for m in gmodes}
    for _ in range(args.rounds):
        for variant, fma in gmodes:
            ctx.set_option("tune_gaussian_variant", variant)
            ctx.set_option("gaussian_fma_div", fma)
            gres[(variant, fma)].append(time_stage(lambda: ctx.dev_gaussian(d_img, args.sigma, H, W, F, d_sm),
                                                   capi.STAGE_GAUSSIAN))
    ctx.set_option("gaussian_fma_div", 1)
    ctx.set_option("tune_gaussian_variant", 0)
    for variant, fma in gmodes:
        g = gres[(variant, fma)]
        name = {2: "symmetric-tap", 1: "LDS ring", 0: "symmetric-tap + product table"}[variant] + \
            (", fma division" if fma else ", 5-op division")
        print(f"gaussian march [{name}]  median {statistics.median(g):.4f} ms  min {min(g):.4f} ms")
    # worst case for the product table's LDS bank conflicts: uniformly random pixels
    rng = np.random.default_rng(7)
    noise = rng.integers(0, 256, size=(H, W), dtype=np.uint8)
    d_noise = ctx.malloc(F * H * W)
    for i in range(F):
        ctx.h2d(d_noise + i * H * W, np.roll(noise, 977 * i))
    nres = {0: [], 2: []}
    for _ in range(args.rounds):
        for variant in (0, 2):
            ctx.set_option("tune_gaussian_variant", variant)
            nres[variant].append(time_stage(lambda: ctx.dev_gaussian(d_noise, args.sigma, H, W, F, d_sm),
                                            capi.STAGE_GAUSSIAN))
    ctx.set_option("tune_gaussian_variant", 0)
    ctx.free(d_noise)
    for variant in (0, 2):
        print(f"gaussian on uniform noise, variant {variant}: median {statistics.median(nres[variant]):.4f} ms  "
              f"min {min(nres[variant]):.4f} ms")
    segs = (0, 64, 100, 188, 265, 441)
    sres = {s: [] for s in segs}
    for _ in range(args.rounds):
        for s in segs:
            ctx.set_option("tune_gaussian_seg", s)
            sres[s].append(time_stage(lambda: ctx.dev_gaussian(d_img, args.sigma, H, W, F, d_sm), capi.STAGE_GAUSSIAN))
    ctx.set_option("tune_gaussian_seg", 0)
    for s in segs:
        print(f"gaussian symmetric-tap, segment target {s or 'auto':>4}: median {statistics.median(sres[s]):.4f} ms  "
              f"min {min(sres[s]):.4f} ms")
    ctx.dev_sobel_nms(d_sm, H, W, F, d_out)
    hy = {}
    for _ in range(args.rounds):
        for fmode in (0, 1):
            ctx.set_option("tune_finalize_mode", fmode)
            ctx.dev_sobel_nms(d_sm, H, W, F, d_out)
            ctx.profile_reset()
            ctx.dev_hysteresis(d_out, H, W, F, 50, 150)
            for k, sid in (("hyst_classify", 2), ("hyst_propagate", 3), ("hyst_finalize", 4)):
                ms, n = ctx.profile_get(sid)
                hy.setdefault(f"{k}[finalize_mode={fmode}]", []).append(ms)
    ctx.set_option("tune_finalize_mode", 0)
    for k, v in hy.items():
        print(f"{k:36s} median {statistics.median(v):.4f} ms  min {min(v):.4f} ms")
    print("hysteresis sweeps:", ctx.last_hysteresis_iterations)

    # whole pipeline, with and without the Sobel+NMS kernel writing the bit-planes itself
    import time
    names = ["gaussian", "sobel_nms", "hyst_classify", "hyst_propagate", "hyst_finalize"]
    pipe = {}
    modes = ((0, 0), (1, 0), (1, 1))  # (fuse_classify, overlap_hysteresis)
    for _ in range(args.rounds):
        for fuse, overlap in modes:
            ctx.set_option("fuse_classify", fuse)
            ctx.set_option("overlap_hysteresis", overlap)
            ctx.profile_reset()
            ctx.synchronize()
            t0 = time.perf_counter()
            ctx.dev_canny(d_img, args.sigma, 50, 150, H, W, F, d_out)
            ctx.synchronize()
            wall = (time.perf_counter() - t0) * 1e3
            row = [ctx.profile_get(sid)[0] for sid in range(5)]
            pipe.setdefault((fuse, overlap), []).append(row + [wall])
    ctx.set_option("fuse_classify", 1)
    ctx.set_option("overlap_hysteresis", 0)
    for fuse, overlap in modes:
        med = [statistics.median(col) for col in zip(*pipe[(fuse, overlap)])]
        parts = "  ".join(f"{n} {v:.3f}" for n, v in zip(names, med))
        print(f"canny fuse_classify={fuse} overlap_hysteresis={overlap}: {parts}  | stages {sum(med[:5]):.3f} ms, "
              f"wall {med[5]:.3f} ms")
    ctx.close()


if __name__ == "__main__":
    main()
