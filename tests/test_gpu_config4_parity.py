"""BASELINE config 4 at FULL size -- one 16384 x 16384 frame, sigma 2.0, thresholds 50/150 -- with the FINAL EDGE MAP
compared with the oracle pixel for pixel (round 2 compared a band of the smoothed and suppressed planes only), and
8192 x 8192 as the cheaper sibling.

The frame is built so that hysteresis has real work on the 256 x 256 tile grid (src/utils.cpp:322-427): faint straight
edges that cross the WHOLE image (horizontal ones in the upper half, vertical ones in the lower) -- weak after smoothing, i.e. between the thresholds -- each seeded by
one short high-contrast stretch, so almost every edge pixel of the result is a PROMOTED pixel and a promotion travels up
to ~250 tiles from its seed.  Frames of more than 4096 tiles do not take the one-workgroup-per-frame tail kernel: this
is the multi-launch propagation (device work queue, host convergence poll) on a 2-D grid.  Nothing here is periodic:
positions, seeds and the block clutter come from a seeded generator.

CPU cost on the GPU box's host: ~15 s of oracle for 16384^2 plus ~10 s to build the frame."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

LO, HI, SIGMA = 50, 150, 2.0


def long_edge_frame(n, seed):
    """u8 frame n x n, background 30.  Upper half: ~3 n / 1024 faint straight bars (14 px wide, 45 above the background:
    Sobel magnitude ~70 on either side after the sigma-2 blur, between the thresholds) that run across the whole WIDTH;
    lower half: as many that run down the whole half.  Edges never cross and never change rows (a crossing cuts both edges -- the gradient turns and the
    non-maximum test drops the corner pixels --, and where a slanted edge steps to the next row its two rows tie and the
    strict test drops both), so a promotion walks the full length of its edge.  Every edge carries
    ONE ~60-pixel stretch of step 200 (magnitude ~320: strong) at a random place.  The edge profile is 0, 1/3, 1 of the
    step -- a symmetric step leaves two equal magnitudes side by side and the reference's strict non-maximum test
    suppresses both (tests/test_gpu_promotion_patterns.py).  Plus a few dozen filled blocks (strong borders that cut an
    edge here and there) and +-1 noise (more than that ties the two rows of a faint edge here and there, and a tie cuts it)."""
    rng = np.random.default_rng(seed)
    n_lines = max(3, 3 * n // 1024)
    half = n // 2
    ys = np.sort(rng.choice(np.arange(60, half - 60, 97), n_lines, replace=False)).astype(np.float32)
    xs = np.sort(rng.choice(np.arange(40, n - 40, 41), n_lines, replace=False)).astype(np.float32)
    seed_x = rng.uniform(100, n - 100, n_lines).astype(np.float32)
    seed_y = rng.uniform(half + 120, n - 100, n_lines).astype(np.float32)
    out = np.empty((n, n), np.uint8)
    xx = np.arange(n, dtype=np.float32)[None, :]

    def edge(dist, height):  # a bar 14 pixels wide: its two sides are two parallel edges
        up = np.where(dist >= 1, 1.0, np.where(dist >= 0, 1.0 / 3.0, 0.0))
        down = np.where(dist >= 15, 1.0, np.where(dist >= 14, 1.0 / 3.0, 0.0))
        return ((up - down) * height).astype(np.float32)

    def bump(along, centre):  # 45 everywhere, rising smoothly to 200 around the seed: the edge direction never changes
        return 45.0 + 155.0 * np.exp(-((along - centre) / 20.0) ** 2)

    B = 512
    for y0 in range(0, n, B):
        yy = np.arange(y0, min(n, y0 + B), dtype=np.float32)[:, None]
        acc = np.full((yy.shape[0], n), 30.0, np.float32)
        if y0 < half:
            for k in range(n_lines):
                if abs(ys[k] - (y0 + B / 2)) < B / 2 + 40:
                    acc += edge(yy - ys[k], 1.0) * bump(xx, seed_x[k])
        if y0 + B > half + 20:
            below = (yy >= half + 20).astype(np.float32)
            for k in range(n_lines):
                acc += edge(xx - xs[k], 1.0) * bump(yy, seed_y[k]) * below
        out[y0:y0 + B] = np.clip(np.rint(acc), 0, 255).astype(np.uint8)
    for _ in range(max(4, n // 512)):  # clutter: filled blocks of random grey
        y, x = int(rng.integers(0, n - 8)), int(rng.integers(0, n - 8))
        out[y:y + int(rng.integers(8, 120)), x:x + int(rng.integers(8, 120))] = rng.integers(0, 256)
    noise = rng.integers(-1, 2, (n, n), dtype=np.int8)
    return np.clip(out.astype(np.int16) + noise, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("n", [8192, 16384], ids=["8192x8192", "16384x16384_BASELINE_config_4"])
def test_config4_final_edge_map_equals_oracle(hip, n):
    img = long_edge_frame(n, seed=4000 + n)
    ref = oracle.canny(img, SIGMA, LO, HI, stages=True)
    want, nms = ref["edges"], ref["nms"]
    # the frame really is a propagation workload: most edge pixels are promoted ones, and they lie far from any seed
    strong = int(np.count_nonzero(nms >= HI))
    promoted = int(np.count_nonzero((want == 255) & (nms < HI)))
    assert promoted > 2 * n * (3 * n // 1024) // 2 and promoted > 3 * strong, (promoted, strong)
    with hip.Context(0) as c:
        got = c.canny(img, SIGMA, LO, HI)
        sweeps = c.last_hysteresis_iterations
        bad = np.argwhere(got != want)
        assert bad.size == 0, (n, len(bad), bad[:5].tolist())
        assert sweeps > 60, sweeps                     # promotions really walked tile by tile
        # device-resident call on the same frame, twice (workspaces re-used), and the stage API's hysteresis on the
        # oracle's suppressed plane: the same map again
        d_in, d_out = c.malloc(img.nbytes), c.malloc(img.nbytes * 2)
        try:
            c.h2d(d_in, img)
            for _ in range(2):
                c.dev_canny(d_in, SIGMA, LO, HI, n, n, 1, d_out)
                again = np.empty((n, n), np.int16)
                c.d2h(again, d_out)
                assert np.array_equal(again, want)
        finally:
            c.free(d_in)
            c.free(d_out)
        assert np.array_equal(c.hysteresis(nms, LO, HI), want)
