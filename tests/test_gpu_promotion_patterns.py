"""canny() on frames whose edge maps are mostly PROMOTED pixels (weak pixels connected to a small strong seed), in the
shapes that take different routes through the propagation kernels: a tile's promoted pixels are written to the edge map
one store per changed row (lane = column) when at most 16 of its rows changed and one row per lane otherwise
(canny_kernels.hip, process_tile_with), and the hand-over from the batch-wide sweeps to the per-frame tail kernel
happens after two sweeps -- so long horizontal runs, long vertical runs and a grid of both, each many tiles long, all have to come out as the reference's flood does (src/utils.cpp:360-427), frame by frame and in batches."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

H, W = 520, 780  # 9 x 13 tiles of 64 x 64


def _step_frames():
    """Faint edges (weak after smoothing) with one short high-contrast stretch (strong) to seed them.  The edge profile
    is 0, 1/3, 1 of the step: a symmetric step gives two equal magnitudes side by side and the reference's strict
    non-maximum test (src/utils.cpp:248-308) suppresses both."""
    frames = {}
    yy, xx = np.mgrid[0:H, 0:W]
    base = np.full((H, W), 100.0)

    def edge(dist, height):  # dist: signed distance from the edge line, in pixels
        return np.where(dist >= 1, height, np.where(dist >= 0, height / 3.0, 0.0))

    def bump(along, centre):  # the edge's height along its length: 10 everywhere, rising smoothly to 110 at the seed,
        return 10.0 + 100.0 * np.exp(-((along - centre) / 12.0) ** 2)  # so that the edge direction never changes

    # horizontal edge along y = 200: few rows, runs as long as the frame is wide
    frames["horizontal"] = base + edge(yy - 200, 1.0) * bump(xx, 30)
    # vertical edge along x = 301: every row of a tile column changes, one or two pixels per row
    frames["vertical"] = base + edge(xx - 301, 1.0) * bump(yy, 40)
    # both at once, several of each: tiles that are entered from more than one side in different sweeps
    both = base.copy()
    for k, y0 in enumerate((70, 250, 431)):
        both = both + edge(yy - y0, 1.0) * bump(xx, 60 + 250 * k)
    for k, x0 in enumerate((130, 390, 649)):
        both = both + edge(xx - x0, 1.0) * bump(yy, 480 - 200 * k)
    frames["grid"] = both
    # (45-degree edges are no use here: the reference's non-maximum test for the diagonal bins compares ALONG such an
    # edge, src/utils.cpp:248-308, so only isolated pixels of it survive -- nothing to promote)
    return {k: np.clip(np.rint(v), 0, 255).astype(np.uint8) for k, v in frames.items()}


@pytest.fixture(scope="module")
def cases():
    out = {}
    for name, img in _step_frames().items():
        r = oracle.canny(img, 1.0, 12, 120, stages=True)
        edges, nms = r["edges"], r["nms"]
        promoted = int(np.count_nonzero((edges == 255) & (nms < 120)))
        strong = int(np.count_nonzero(nms >= 120))
        out[name] = (img, edges, promoted, strong)
    return out


@pytest.mark.parametrize("name", ["horizontal", "vertical", "grid"])
def test_promoted_runs_across_many_tiles(hip, cases, name):
    img, want, promoted, strong = cases[name]
    # the case is what it claims to be: far more promoted than seed pixels, spread over many tiles
    assert promoted > 10 * max(1, strong) and promoted > 400, (name, promoted, strong)
    tiles = {(y // 64, x // 64) for y, x in zip(*np.nonzero(want))}
    assert len(tiles) >= 9, (name, len(tiles))
    with hip.Context(0) as ctx:
        for tail in (1, 0):
            ctx.set_option("hysteresis_tail", tail)
            assert np.array_equal(ctx.canny(img, 1.0, 12, 120), want), (name, tail)
        ctx.set_option("hysteresis_tail", 1)
        for after in (1, 2, 3):
            ctx.set_option("tune_hyst_tail_after", after)
            assert np.array_equal(ctx.canny(img, 1.0, 12, 120), want), (name, "after", after)


def test_promotion_patterns_in_one_batch(hip, cases):
    """All three in one batch (each frame has its own queues in the tail kernel), twice over, mirrored and transposed."""
    imgs, wants = [], []
    for name in ("horizontal", "vertical", "grid"):
        img = cases[name][0]
        for variant in (img, img[::-1].copy(), img[:, ::-1].copy()):
            imgs.append(variant)
            wants.append(oracle.canny(variant, 1.0, 12, 120))
    frames, want = np.stack(imgs), np.stack(wants)
    with hip.Context(0) as ctx:
        got = ctx.canny_batch(frames, 1.0, 12, 120)
        assert np.array_equal(got, want)
        assert np.array_equal(hip.unpack_bits(ctx.canny_batch(frames, 1.0, 12, 120, bits=True), W), want)
