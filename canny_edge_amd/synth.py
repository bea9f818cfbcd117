"""Deterministic synthetic grayscale frames for tests and benchmarks.

The reference has no frame source other than a webcam (``src/main.cpp:78-115``), so benchmarks
and parity tests use this generator instead: a flat background, a few hundred filled rectangles of
random gray level, and +-8 uniform noise.  That gives every stage real work (edges for Sobel/NMS,
weak and strong pixels for hysteresis); smooth low-frequency images would produce no edges at all.
"""
from __future__ import annotations

import numpy as np


def synth_frame(height: int, width: int, seed: int = 42) -> np.ndarray:
    """Return a uint8 [height, width] frame; the same (shape, seed) always gives the same bytes.
    The generator is the Mersenne Twister SURVEY.md 8(d) names (mt19937, seed 42 + i for frame i)."""
    rng = np.random.Generator(np.random.MT19937(seed))
    img = np.full((height, width), 30, dtype=np.int16)
    n_rect = (height * width) // 8000 + 4
    max_w = max(9, width // 6)
    max_h = max(9, height // 6)
    xs = rng.integers(0, max(1, width), size=n_rect)
    ys = rng.integers(0, max(1, height), size=n_rect)
    ws = rng.integers(8, max_w + 1, size=n_rect)
    hs = rng.integers(8, max_h + 1, size=n_rect)
    levels = rng.integers(0, 256, size=n_rect)
    for x0, y0, w, h, lv in zip(xs, ys, ws, hs, levels):
        img[y0:y0 + h, x0:x0 + w] = lv
    img += rng.integers(-8, 9, size=(height, width), dtype=np.int16)
    return np.clip(img, 0, 255).astype(np.uint8)


def synth_batch(n_frames: int, height: int, width: int, seed: int = 42, distinct: int = 16) -> np.ndarray:
    """Return uint8 [n_frames, height, width]; frame i uses seed + (i % distinct)."""
    distinct = max(1, min(distinct, n_frames))
    base = [synth_frame(height, width, seed + i) for i in range(distinct)]
    out = np.empty((n_frames, height, width), dtype=np.uint8)
    for i in range(n_frames):
        out[i] = base[i % distinct]
    return out
