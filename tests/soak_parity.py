#!/usr/bin/env python3
"""Randomised parity soak: the HIP path against the oracle on random shapes, pictures, sigmas, thresholds, batch sizes
and tuning options, for a bounded time (default 300 s).  The -m gpu tests fix their cases; this draws new ones -- the
point is the rare interleaving (the per-frame tail kernel, the three-slot batch pipeline, context reuse across
shapes), not coverage of a feature.  Not collected by pytest (no test_ prefix): run it by hand on a GPU box.

    python tests/soak_parity.py [--seconds 300] [--seed N] [--max-pixels 1500000]

Prints one line per 50 cases and a JSON summary; exits 1 on the first mismatch, with the case's parameters and seed.
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def picture(rng, kind, h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == "noise":
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    if kind == "flat":
        return np.full((h, w), int(rng.integers(0, 256)), np.uint8)
    if kind == "blocks":
        img = np.full((h, w), 30, np.int32)
        for _ in range(int(rng.integers(1, 40))):
            y0, x0 = int(rng.integers(0, h)), int(rng.integers(0, w))
            img[y0:y0 + int(rng.integers(1, max(2, h // 3))), x0:x0 + int(rng.integers(1, max(2, w // 3)))] = rng.integers(0, 256)
        return np.clip(img + rng.integers(-8, 9, (h, w)), 0, 255).astype(np.uint8)
    if kind == "stripes":  # long thin edges: many tile crossings for the hysteresis
        period = int(rng.integers(3, 40))
        a = ((xx + (yy // max(1, int(rng.integers(1, 9))))) % period < period // 2) * int(rng.integers(40, 256))
        return np.clip(a + rng.integers(-5, 6, (h, w)), 0, 255).astype(np.uint8)
    if kind == "spiral":  # one connected weak curve seeded at a single strong spot
        cy, cx = h / 2.0, w / 2.0
        r = np.hypot(yy - cy, xx - cx)
        th = np.arctan2(yy - cy, xx - cx)
        img = 20 + 60 * (np.sin(r / 3.0 - 2 * th) > 0.6)
        img[int(cy) - 1:int(cy) + 2, int(cx) - 1:int(cx) + 2] = 255
        return img.astype(np.uint8)
    grad = (xx * int(rng.integers(1, 5)) + yy * int(rng.integers(1, 5))) % 256  # "gradient": sawtooth ramps
    return np.clip(grad + rng.integers(-20, 21, (h, w)), 0, 255).astype(np.uint8)


KINDS = ["noise", "flat", "blocks", "blocks", "stripes", "spiral", "gradient"]
SIGMAS = [0.3, 0.5, 0.8, 1.0, 1.0, 1.4, 1.4, 2.0, 2.5, 3.3, 5.0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=int(time.time()))
    ap.add_argument("--max-pixels", type=int, default=1_500_000)
    args = ap.parse_args()

    import oracle
    from canny_edge_amd import capi

    rng = np.random.default_rng(args.seed)
    pool = ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4))  # the oracle is C behind ctypes: no GIL held
    ctx = capi.Context(0)
    t0 = time.time()
    cases = frames_done = pixels = 0
    by_path = {}
    while time.time() - t0 < args.seconds:
        h = int(rng.choice([int(rng.integers(2, 12)), int(rng.integers(2, 200)), int(rng.integers(2, 1300))]))
        w = int(rng.choice([int(rng.integers(2, 12)), int(rng.integers(2, 300)), int(rng.integers(2, 2100))]))
        while h * w > args.max_pixels:
            h = max(2, h // 2)
        n = int(rng.choice([1, 1, 2, 3, 5, 9]))
        while n > 1 and n * h * w > 2 * args.max_pixels:
            n -= 1
        kind = KINDS[int(rng.integers(0, len(KINDS)))]
        sigma = float(SIGMAS[int(rng.integers(0, len(SIGMAS)))])
        lo, hi = int(rng.integers(0, 200)), int(rng.integers(1, 256))
        mode = int(rng.integers(0, 10))
        if mode >= 3 and lo >= hi:          # mostly lo < hi like the CLI enforces; sometimes anything goes
            lo, hi = min(lo, hi), max(lo, hi) + 1
        frames = np.stack([picture(rng, kind, h, w) for _ in range(n)])
        want = list(pool.map(lambda f: oracle.canny(f, sigma, lo, hi), frames))
        opts = {"hysteresis_tail": int(rng.integers(0, 2)) if rng.random() < 0.3 else 1,
                "smoothed_u8": int(rng.random() < 0.6),
                "tune_batch_chunk_frames": int(rng.integers(0, 4)),
                "tune_batch_pipe_mode": int(rng.integers(0, 3)),
                "tune_batch_workers": int(rng.integers(0, 4))}
        for k, v in opts.items():
            ctx.set_option(k, v)
        path = int(rng.integers(0, 5))
        name = ("canny per frame", "canny_batch s16", "canny_batch u8", "stage calls", "canny_batch bits")[path]
        what = dict(seed=args.seed, case=cases, h=h, w=w, n=n, kind=kind, sigma=sigma, lo=lo, hi=hi, path=name, **opts)
        try:
            if path == 0:
                got = [ctx.canny(f, sigma, lo, hi) for f in frames]
            elif path == 1:
                got = list(ctx.canny_batch(frames, sigma, lo, hi))
            elif path == 2:
                got = [g.astype(np.int16) for g in ctx.canny_batch(frames, sigma, lo, hi, u8=True)]
            elif path == 4:
                got = list(capi.unpack_bits(ctx.canny_batch(frames, sigma, lo, hi, bits=True), w))
            else:
                got = []
                for f in frames:
                    mag, ang = ctx.sobel(ctx.gaussian(f, sigma))
                    got.append(ctx.hysteresis(ctx.nms(mag, ang), lo, hi))
        except capi.CannyHipError as e:
            # thresholds whose result depends on the reference's scan order are refused by design:
            # min > 255 >= max.  Nothing else may fail.
            if not (lo > 255 >= hi):
                print("UNEXPECTED ERROR", e, json.dumps(what))
                sys.exit(1)
            continue
        for i in range(n):
            if not np.array_equal(got[i], want[i]):
                what["frame"] = i
                what["mismatching_pixels"] = int((got[i] != want[i]).sum())
                print("MISMATCH", json.dumps(what))
                sys.exit(1)
        cases += 1
        frames_done += n
        pixels += n * h * w
        by_path[name] = by_path.get(name, 0) + 1
        if cases % 50 == 0:
            print(f"[soak] {cases} cases, {frames_done} frames, {pixels / 1e6:.0f} Mpix, {time.time() - t0:.0f} s", flush=True)
    ctx.close()
    print(json.dumps({"soak": "ok", "seed": args.seed, "cases": cases, "frames": frames_done,
                      "megapixels": round(pixels / 1e6, 1), "seconds": round(time.time() - t0, 1), "by_path": by_path}))


if __name__ == "__main__":
    main()
