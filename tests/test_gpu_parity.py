"""Parity tests proper: the HIP path, called through the C ABI, against the CPU oracle and against
the reference's own known-answer vectors.  Integer planes must be BIT-EXACT; there is no tolerance
anywhere in this file (the pipeline has no floating-point output: the float Gaussian intermediate
is truncated to short inside the kernel, exactly like the reference)."""
import hashlib

import numpy as np
import pytest

import oracle
from canny_edge_amd.synth import synth_frame

pytestmark = pytest.mark.gpu


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def ctx(hip):
    c = hip.Context(0)
    yield c
    c.close()


def _noise(h, w, seed):
    return np.random.default_rng(seed).integers(0, 256, size=(h, w), dtype=np.uint8)


def _mixed(h, w, seed):
    """Blocks + gradients + noise: many weak/strong pixels and long connected edges."""
    img = synth_frame(h, w, seed).astype(np.int16)
    yy, xx = np.mgrid[0:h, 0:w]
    img = img // 2 + ((xx * 3 + yy * 2) % 128).astype(np.int16)
    return np.clip(img, 0, 255).astype(np.uint8)


SIZES = [(2, 2), (2, 9), (9, 2), (3, 3), (5, 7), (16, 16), (31, 33), (64, 64), (65, 63), (97, 131), (128, 192),
         (200, 257), (240, 320)]


# ---------------------------------------------------------------------------------------------
# 1. The reference's own vectors replayed through the HIP C-ABI (tests/utils/test_utils.cpp)
# ---------------------------------------------------------------------------------------------
def test_ref_gradient_vectors(hip, ref_vectors):
    for case in ref_vectors["gradient"]:
        img = np.array(case["img"], np.int16).reshape(case["rows"], case["columns"])
        gx, gy = hip.calculateXYGradient(img)
        assert gx.ravel().tolist() == case["gx"], case["name"]
        assert gy.ravel().tolist() == case["gy"], case["name"]


def test_ref_sobel_constant(hip, ref_vectors):
    case = ref_vectors["sobel"][0]
    img = np.array(case["img"], np.int16).reshape(3, 3)
    mag, ang = hip.sobelOperator(img)
    assert mag.shape == ang.shape == (3, 3) and not mag.any() and not ang.any()


def test_ref_angle_vector(ctx, ref_vectors):
    case = ref_vectors["angle_bins"][0]
    _, bins = ctx.selftest_mag_angle(8)
    got = [int(bins[gy + 8, gx + 8]) for gx, gy in zip(case["gx"], case["gy"])]
    assert got == case["angle"]


def test_ref_nms_vectors(hip, ref_vectors):
    for case in ref_vectors["nms"]:
        shape = (case["rows"], case["columns"])
        out = hip.nonmaximalSuppression(np.array(case["grad"], np.int16).reshape(shape),
                                        np.array(case["angle"], np.int16).reshape(shape))
        assert out.ravel().tolist() == case["expected"], case["name"]


def test_ref_find_edge_pixels_vector(hip, ref_vectors):
    case = ref_vectors["find_edge_pixels"][0]
    shape = (case["rows"], case["columns"])
    out, _ = hip.findEdgePixels(np.array(case["suppress"], np.int16).reshape(shape), np.zeros(shape, np.uint8),
                                case["start"], case["min"], case["max"])
    assert out.ravel().tolist() == case["expected"]


def test_ref_hysteresis_vector(hip, ref_vectors):
    case = ref_vectors["hysteresis"][0]
    shape = (case["rows"], case["columns"])
    out = hip.hysteresis(np.array(case["suppress"], np.int16).reshape(shape), case["min"], case["max"])
    assert out.ravel().tolist() == case["expected"]


def test_ref_gaussian_fixture(hip, ref_vectors, fixture_image):
    case = ref_vectors["gaussian_image"][0]
    out = hip.gaussian(fixture_image, case["sigma"])
    assert out.shape == (256, 256) and int(out.astype(np.int64).sum()) != 0
    assert out.min() >= 0 and out.max() <= 255


# ---------------------------------------------------------------------------------------------
# 2. Device arithmetic rules, exhaustively
# ---------------------------------------------------------------------------------------------
def test_device_magnitude_and_angle_exhaustive(ctx):
    """Every gradient a [0,255] plane can produce: |gx|,|gy| <= 1020 (4.16 M pairs)."""
    mags, bins = ctx.selftest_mag_angle(1020)
    assert np.array_equal(bins, oracle.angle_table(1020))
    assert np.array_equal(mags, oracle.magnitude_table(1020))


# ---------------------------------------------------------------------------------------------
# 3. Stage-by-stage parity against the oracle on seeded inputs
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("sigma", [0.3, 0.5, 1.0, 1.4, 2.0, 2.7, 4.0])
@pytest.mark.parametrize("shape", [(1, 1), (1, 40), (40, 1), (3, 5), (17, 19), (64, 64), (97, 131), (130, 260)])
def test_gaussian_parity(ctx, sigma, shape):
    for seed, gen in ((1, _noise), (2, _mixed)):
        img = gen(shape[0], shape[1], seed)
        assert np.array_equal(ctx.gaussian(img, sigma), oracle.gaussian(img, sigma)), (sigma, shape, seed)


def test_gaussian_extremes(ctx):
    for fill in (0, 255):
        img = np.full((50, 70), fill, np.uint8)
        assert np.array_equal(ctx.gaussian(img, 1.4), oracle.gaussian(img, 1.4))
    img = np.zeros((40, 40), np.uint8)
    img[::2, ::2] = 255
    for s in (0.5, 1.0, 2.0, 6.5):
        assert np.array_equal(ctx.gaussian(img, s), oracle.gaussian(img, s))


@pytest.mark.parametrize("shape", SIZES)
def test_gradient_sobel_parity(ctx, shape):
    h, w = shape
    sm = oracle.gaussian(_mixed(h, w, 3), 1.0)
    for plane in (sm, _noise(h, w, 4).astype(np.int16)):
        gx, gy = ctx.xy_gradient(plane)
        ogx, ogy = oracle.xy_gradient(plane)
        assert np.array_equal(gx, ogx) and np.array_equal(gy, ogy)
        mag, ang = ctx.sobel(plane)
        omag, oang = oracle.sobel(plane)
        assert np.array_equal(mag, omag) and np.array_equal(ang, oang)


def test_sobel_general_domain_wraps_like_the_reference(ctx):
    """Arbitrary shorts: gradients wrap through short, magnitude through (short)(int)sqrt."""
    rng = np.random.default_rng(5)
    plane = rng.integers(-3000, 3000, size=(33, 47), dtype=np.int16)
    gx, gy = ctx.xy_gradient(plane)
    ogx, ogy = oracle.xy_gradient(plane)
    assert np.array_equal(gx, ogx) and np.array_equal(gy, ogy)
    mag, _ = ctx.sobel(plane)
    assert np.array_equal(mag, oracle.sobel(plane)[0])


@pytest.mark.parametrize("shape", SIZES + [(1, 1), (1, 9), (9, 1)])
def test_nms_parity(ctx, shape):
    h, w = shape
    rng = np.random.default_rng(6)
    mag = rng.integers(0, 40, size=(h, w), dtype=np.int16)          # many ties
    ang = rng.choice(np.array([0, 45, 90, 135], np.int16), size=(h, w))
    assert np.array_equal(ctx.nms(mag, ang), oracle.nms(mag, ang))
    mag2 = rng.integers(0, 1443, size=(h, w), dtype=np.int16)
    assert np.array_equal(ctx.nms(mag2, ang), oracle.nms(mag2, ang))


@pytest.mark.parametrize("shape", SIZES)
def test_fused_sobel_nms_equals_separate_stages(ctx, hip, shape):
    h, w = shape
    for seed in (7, 8):
        sm = oracle.gaussian(_mixed(h, w, seed), 1.0)
        want = oracle.nms(*oracle.sobel(sm))
        d_in, d_out = ctx.malloc(sm.nbytes), ctx.malloc(sm.nbytes)
        try:
            ctx.h2d(d_in, sm)
            ctx.dev_sobel_nms(d_in, h, w, 1, d_out)
            got = np.empty_like(sm)
            ctx.d2h(got, d_out)
        finally:
            ctx.free(d_in)
            ctx.free(d_out)
        assert np.array_equal(got, want), (shape, seed)


@pytest.mark.parametrize("shape", SIZES + [(1, 1), (1, 70), (70, 1), (63, 129), (129, 65)])
@pytest.mark.parametrize("lohi", [(50, 150), (1, 2), (10, 10), (0, 1), (100, 300), (200, 100)])
def test_hysteresis_parity(ctx, shape, lohi):
    h, w = shape
    lo, hi = lohi
    rng = np.random.default_rng(9)
    for density in (0.05, 0.3, 0.9):
        cand = rng.integers(0, 256, size=(h, w), dtype=np.int16)
        cand[rng.random((h, w)) > density] = 0
        assert np.array_equal(ctx.hysteresis(cand, lo, hi), oracle.hysteresis(cand, lo, hi)), (shape, lohi, density)


def test_hysteresis_directed_quirk(ctx):
    """(1,0) never pushes (0,1) (src/utils.cpp:378,399) but (0,1) pushes (1,0)."""
    a = np.zeros((4, 4), np.int16)
    a[2, 0], a[1, 0], a[0, 1] = 200, 60, 60
    out = ctx.hysteresis(a, 50, 150)
    assert np.array_equal(out, oracle.hysteresis(a, 50, 150)) and out[0, 1] == 0 and out[1, 0] == 255
    b = np.zeros((4, 4), np.int16)
    b[0, 2], b[0, 1], b[1, 0] = 200, 60, 60
    out = ctx.hysteresis(b, 50, 150)
    assert np.array_equal(out, oracle.hysteresis(b, 50, 150)) and out[1, 0] == 255
    # the quirk pixel pair inside a larger image and with other ways round it
    for seed in range(20):
        rng = np.random.default_rng(100 + seed)
        c = (rng.random((6, 6)) < 0.45).astype(np.int16) * 60
        c[rng.integers(0, 6), rng.integers(0, 6)] = 200
        assert np.array_equal(ctx.hysteresis(c, 50, 150), oracle.hysteresis(c, 50, 150)), seed


def test_hysteresis_long_paths_cross_many_tiles(ctx):
    """A serpentine one-pixel path through a 200x330 image: thousands of steps, dozens of 64x64 tiles."""
    h, w = 200, 330
    cand = np.zeros((h, w), np.int16)
    for r in range(0, h, 4):
        cand[r, :] = 60
        if (r // 4) % 2 == 0:
            cand[r:r + 4, w - 1] = 60
        else:
            cand[r:r + 4, 0] = 60
    cand[0, 0] = 250
    want = oracle.hysteresis(cand, 50, 150)
    got = ctx.hysteresis(cand, 50, 150)
    assert np.array_equal(got, want)
    assert np.count_nonzero(got) == np.count_nonzero(cand)     # everything is reached
    assert ctx.last_hysteresis_iterations > 4
    # break the path: nothing beyond the gap may be reached
    cand[100, 150] = 0
    assert np.array_equal(ctx.hysteresis(cand, 50, 150), oracle.hysteresis(cand, 50, 150))


def test_hysteresis_rejects_order_dependent_domain(ctx, hip):
    a = np.array([[-5, 10], [3, 0]], np.int16)
    with pytest.raises(hip.CannyHipError) as ei:
        ctx.hysteresis(a, -1, 5)
    assert ei.value.status == 5


@pytest.mark.parametrize("shape", [(5, 5), (20, 31), (64, 64), (70, 130)])
def test_find_edge_pixels_parity(ctx, shape):
    h, w = shape
    rng = np.random.default_rng(11)
    for trial in range(12):
        cand = rng.integers(0, 40, size=(h, w), dtype=np.int16)
        cand[rng.random((h, w)) > 0.55] = 0
        visited = (rng.random((h, w)) < 0.1).astype(np.uint8)
        start = int(rng.integers(0, h * w))
        lo = int(rng.integers(1, 20))
        got_c, got_v = ctx.find_edge_pixels(cand, visited, start, lo, 30)
        want_c, want_v = oracle.find_edge_pixels(cand, visited, start, lo, 30)
        assert np.array_equal(got_c, want_c), (shape, trial)
        assert np.array_equal(got_v, want_v), (shape, trial)


# ---------------------------------------------------------------------------------------------
# 4. Whole pipeline
# ---------------------------------------------------------------------------------------------
def test_pipeline_matches_committed_oracle_hashes(ctx, oracle_hashes, fixture_image, luma_image):
    inputs = {
        "fixture256_s0.5_50_150": fixture_image, "fixture256_s1.0_50_150": fixture_image,
        "jpegluma256_s1.0_50_150": luma_image,
        "synth_97x131_seed7_s1.4_50_150": synth_frame(97, 131, 7),
        "synth_240x320_seed42_s2.0_30_90": synth_frame(240, 320, 42),
        "synth_64x64_seed3_s0.5_10_50": synth_frame(64, 64, 3),
    }
    for name, want in oracle_hashes.items():
        img = inputs[name]
        sm = ctx.gaussian(img, want["sigma"])
        assert _sha(sm) == want["smoothed_sha256"], name
        mag, ang = ctx.sobel(sm)
        assert _sha(mag) == want["magnitude_sha256"] and _sha(ang) == want["angle_sha256"], name
        nm = ctx.nms(mag, ang)
        assert _sha(nm) == want["nms_sha256"], name
        assert _sha(ctx.hysteresis(nm, want["min"], want["max"])) == want["edges_sha256"], name
        assert _sha(ctx.canny(img, want["sigma"], want["min"], want["max"])) == want["edges_sha256"], name


@pytest.mark.parametrize("shape,sigma", [((2, 2), 1.0), ((37, 53), 0.5), ((97, 131), 1.4), ((256, 256), 1.0),
                                         ((300, 500), 2.0), ((480, 640), 0.5), ((1080, 1920), 1.0)])
def test_canny_end_to_end_parity(ctx, shape, sigma):
    img = _mixed(shape[0], shape[1], 21)
    assert np.array_equal(ctx.canny(img, sigma, 50, 150), oracle.canny(img, sigma, 50, 150))


def test_canny_4k_full_size_parity_and_properties(ctx):
    """BASELINE config 2 at full size (3840x2160, sigma 1.4): every stage compared with the oracle
    (about 1.5 s of CPU), plus size-independent properties."""
    h, w = 2160, 3840
    img = synth_frame(h, w, 42)
    ref = oracle.canny(img, 1.4, 50, 150, stages=True)
    sm = ctx.gaussian(img, 1.4)
    assert np.array_equal(sm, ref["smoothed"])
    d_in, d_out = ctx.malloc(sm.nbytes), ctx.malloc(sm.nbytes)
    try:
        ctx.h2d(d_in, sm)
        ctx.dev_sobel_nms(d_in, h, w, 1, d_out)
        nm = np.empty_like(sm)
        ctx.d2h(nm, d_out)
    finally:
        ctx.free(d_in)
        ctx.free(d_out)
    assert np.array_equal(nm, ref["nms"])
    edges = ctx.canny(img, 1.4, 50, 150)
    assert np.array_equal(edges, ref["edges"])
    assert set(np.unique(edges)) <= {0, 255} and 0 < np.count_nonzero(edges) < edges.size // 10
    # idempotence: a finished edge map is a fixed point of hysteresis
    assert np.array_equal(ctx.hysteresis(edges, 50, 150), edges)
    # monotonicity in the strong threshold: raising max can only remove edge pixels
    fewer = ctx.hysteresis(nm, 50, 200)
    assert np.all(edges[fewer == 255] == 255)


def test_canny_16k_tile_properties(ctx):
    """BASELINE config 4 (16384x16384, sigma 2.0) is too big for the oracle to finish in seconds, so the
    whole frame is checked through properties and a 700-row band is compared with the oracle exactly
    (the band's interior is independent of the rest of the image except through hysteresis, so the
    band is compared on the smoothed and NMS planes)."""
    h = w = 16384
    rng = np.random.default_rng(5)
    small = synth_frame(1024, 1024, 9)
    img = np.tile(small, (16, 16))
    img[rng.integers(0, h, 4000), rng.integers(0, w, 4000)] = 255
    edges = ctx.canny(img, 2.0, 50, 150)
    assert edges.shape == (h, w) and set(np.unique(edges)) <= {0, 255}
    assert np.array_equal(ctx.hysteresis(edges, 50, 150), edges)
    band = img[5000:5700]                          # rows 5000..5699
    sm_band = oracle.gaussian(band, 2.0)
    sm_full = ctx.gaussian(img, 2.0)
    assert np.array_equal(sm_full[5006:5694], sm_band[6:694])   # away from the band's own top/bottom border
    nm_band = oracle.nms(*oracle.sobel(sm_full[5000:5700]))
    d_in, d_out = ctx.malloc(sm_full.nbytes), ctx.malloc(sm_full.nbytes)
    try:
        ctx.h2d(d_in, sm_full)
        ctx.dev_sobel_nms(d_in, h, w, 1, d_out)
        nm = np.empty_like(sm_full)
        ctx.d2h(nm, d_out)
    finally:
        ctx.free(d_in)
        ctx.free(d_out)
    assert np.array_equal(nm[5002:5698], nm_band[2:698])


def test_batch_and_multi_gpu_entry_points(ctx, hip):
    frames = np.stack([_mixed(120, 200, 30 + i) for i in range(7)])
    want = np.stack([oracle.canny(f, 1.0, 50, 150) for f in frames])
    assert np.array_equal(ctx.canny_batch(frames, 1.0, 50, 150), want)
    assert np.array_equal(hip.canny_multi_gpu(frames, 1.0, 50, 150, 0), want)
    # device-resident batch: all frames in one launch per stage
    d_in, d_out = ctx.malloc(frames.nbytes), ctx.malloc(frames.nbytes * 2)
    try:
        ctx.h2d(d_in, frames)
        ctx.dev_canny(d_in, 1.0, 50, 150, 120, 200, 7, d_out)
        got = np.empty(frames.shape, np.int16)
        ctx.d2h(got, d_out)
    finally:
        ctx.free(d_in)
        ctx.free(d_out)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("shape", [(120, 200), (33, 37), (64, 1024), (5, 7)])
def test_u8_edge_maps(ctx, shape):
    """The 8-bit edge-map entry points (batch over PCIe and device-resident) carry the same 0 / 255 maps."""
    h, w = shape
    frames = np.stack([_mixed(h, w, 60 + i) for i in range(5)])
    want = np.stack([oracle.canny(f, 1.0, 50, 150) for f in frames])
    assert set(np.unique(want)) <= {0, 255}
    got = ctx.canny_batch(frames, 1.0, 50, 150, u8=True)
    assert got.dtype == np.uint8 and np.array_equal(got.astype(np.int16), want)
    pinned_in = ctx.pinned_array(frames.shape, np.uint8)
    pinned_out = ctx.pinned_array(frames.shape, np.uint8)
    pinned_in[...] = frames
    pinned_out[...] = 7
    assert ctx.canny_batch(pinned_in, 1.0, 50, 150, out=pinned_out, u8=True) is pinned_out
    assert np.array_equal(pinned_out.astype(np.int16), want)
    d_in, d_out = ctx.malloc(frames.nbytes), ctx.malloc(frames.nbytes + 3)
    try:
        ctx.h2d(d_in, frames)
        for shift in (0, 1):  # aligned and unaligned output (the narrowing kernel has a scalar path)
            ctx.dev_canny_u8(d_in, 1.0, 50, 150, h, w, 5, d_out + shift)
            got = np.empty(frames.shape, np.uint8)
            ctx.d2h(got, d_out + shift)
            assert np.array_equal(got.astype(np.int16), want), shift
    finally:
        ctx.free(d_in)
        ctx.free(d_out)


def test_stage_profile_counts_launches(ctx):
    img = _mixed(200, 300, 40)
    ctx.profile_enable(True)
    ctx.profile_reset()
    for _ in range(3):
        ctx.canny(img, 1.0, 50, 150)
    ms, n = ctx.profile_get(1)  # fused sobel+nms
    ctx.profile_enable(False)
    assert n == 3 and ms > 0.0


def test_invalid_arguments_are_rejected(ctx, hip):
    with pytest.raises(hip.CannyHipError):
        ctx.gaussian(np.zeros((4, 4), np.uint8), 0.0)
    with pytest.raises(hip.CannyHipError):
        ctx.sobel(np.zeros((1, 5), np.int16))      # the reference reads out of bounds for H < 2
    with pytest.raises(hip.CannyHipError):
        ctx.find_edge_pixels(np.zeros((3, 3), np.int16), np.zeros((3, 3), np.uint8), 9, 1, 2)


# ---------------------------------------------------------------------------------------------
# 5. Every kernel path (canny_hip_ctx_set_option) gives the same bits
# ---------------------------------------------------------------------------------------------
PATH_SHAPES = [(2, 2), (3, 70), (70, 3), (33, 250), (64, 496), (65, 497), (97, 505), (130, 1000), (300, 1240)]


# path 1 = generic two-pass, 2 = marching (symmetric-tap kernel, systolic row pass, product table in LDS: the
# default), 3 = marching (LDS-ring kernel), 4 = marching (symmetric-tap kernel that multiplies and fetches products),
# 5 = product-fetching row pass with the table (the default of rounds 2-3), 6 = systolic row pass that multiplies;
# the sigmas give half-windows 1..8, i.e. every instantiation of the marching kernels
@pytest.mark.parametrize("path", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("sigma", [0.3, 0.5, 1.0, 1.2, 1.4, 2.0, 2.3, 2.6])
def test_gaussian_paths(hip, path, sigma):
    with hip.Context(0) as c:
        c.set_option("gaussian_path", min(path, 2))
        c.set_option("tune_gaussian_variant", {3: 1, 4: 2, 5: 3, 6: 4}.get(path, 0))
        try:
            for shape in PATH_SHAPES + [(1, 1), (1, 300), (300, 1), (700, 260), (301, 2000)]:
                for seed, gen in ((1, _noise), (2, _mixed)):
                    img = gen(shape[0], shape[1], seed)
                    if path >= 2 and shape[1] < 4:
                        # the marching kernels' border strips load whole dwords inside a row: images narrower than
                        # that belong to the generic kernels (which the automatic choice takes, path 1 above)
                        with pytest.raises(hip.CannyHipError):
                            c.gaussian(img, sigma)
                        continue
                    assert np.array_equal(c.gaussian(img, sigma), oracle.gaussian(img, sigma)), (path, sigma, shape, seed)
        finally:
            c.set_option("tune_gaussian_variant", 0)
        c.set_option("gaussian_path", 0)
        for shape in [(2, 2), (70, 3), (1, 1), (300, 1), (5, 4)]:  # automatic choice on narrow images
            img = _noise(shape[0], shape[1], 3)
            assert np.array_equal(c.gaussian(img, sigma), oracle.gaussian(img, sigma)), (sigma, shape)


def test_gaussian_half_windows_covered():
    centers = sorted({len(oracle.gaussian_kernel(s)) // 2 for s in (0.3, 0.5, 1.0, 1.2, 1.4, 2.0, 2.3, 2.6)})
    assert centers == list(range(1, 9)), centers


@pytest.fixture(params=[(0, 0, 0), (2, 0, 0), (1, 0, 0), (1, 1, 0), (2, 0, 1), (1, 0, 1)],
                ids=["auto", "f32_everywhere", "packed_i16_everywhere", "packed_i16_4px_per_lane",
                     "f32_direct_plane_stores", "packed_i16_direct_plane_stores"])
def sobel_px(hip, request):
    """Runs a test once per variant of the marching Sobel+NMS kernel (process-wide switches): the arithmetic (automatic
    = f32 for the fused classify kernel and packed i16 for the s16 -> s16 kernel, or either one everywhere), pixels per
    lane (packed i16 only), and whether the fused kernel stages its plane bytes in LDS (default) or stores them
    directly."""
    variant, px, direct = request.param
    with hip.Context(0) as c:
        c.set_option("tune_sobel_variant", variant)
        c.set_option("tune_sobel_px", px)
        c.set_option("tune_plane_stores", direct)
    yield request.param
    with hip.Context(0) as c:
        c.set_option("tune_sobel_variant", 0)
        c.set_option("tune_sobel_px", 0)
        c.set_option("tune_plane_stores", 0)


@pytest.mark.parametrize("path", [1, 2])
def test_sobel_nms_paths(hip, path, sobel_px):
    with hip.Context(0) as c:
        c.set_option("sobel_nms_path", path)
        for shape in PATH_SHAPES + [(2, 3), (5, 8), (9, 9), (40, 1489), (600, 130), (70, 2000)]:
            h, w = shape
            for seed in (7, 8, 9):
                sm = oracle.gaussian(_mixed(h, w, seed) if seed != 9 else _noise(h, w, seed), 0.5)
                want = oracle.nms(*oracle.sobel(sm))
                d_in, d_out = c.malloc(sm.nbytes), c.malloc(sm.nbytes)
                try:
                    c.h2d(d_in, sm)
                    c.dev_sobel_nms(d_in, h, w, 1, d_out)
                    got = np.empty_like(sm)
                    c.d2h(got, d_out)
                finally:
                    c.free(d_in)
                    c.free(d_out)
                bad = np.argwhere(got != want)
                assert bad.size == 0, (path, shape, seed, bad[:5].tolist())


# canny() with the Sobel+NMS kernel writing the hysteresis bit-planes itself (fuse_classify=1, the default
# when width % 8 == 0 and min_val >= 1) against the separate classify pass and against the oracle.  Shapes
# cover tile padding in both directions, single-strip and multi-strip widths, one-tile images; thresholds
# cover max < min, equal, unreachable, and min_val = 0 (which must fall back to the unfused kernels).
FUSE_SHAPES = [(2, 8), (3, 16), (64, 64), (65, 72), (130, 496), (100, 504), (63, 520), (129, 1000), (200, 1488),
               (70, 2048)]
FUSE_THRESHOLDS = [(50, 150), (1, 1), (1, 5000), (100, 50), (20, 20), (255, 256), (0, 100), (300, 2000)]


@pytest.mark.parametrize("plane", [1, 0], ids=["u8_smoothed_plane", "s16_smoothed_plane"])
@pytest.mark.parametrize("fuse", [0, 1])
@pytest.mark.parametrize("lo,hi", FUSE_THRESHOLDS)
def test_canny_fused_classify(hip, fuse, lo, hi, sobel_px, plane):
    with hip.Context(0) as c:
        c.set_option("fuse_classify", fuse)
        c.set_option("smoothed_u8", plane)  # 1 = the default: bytes between the Gaussian and the fused kernel
        for h, w in FUSE_SHAPES:
            for n, seed in ((1, 3), (3, 4)):
                frames = np.stack([_mixed(h, w, seed + 10 * i) if i != 1 else _noise(h, w, seed) for i in range(n)])
                want = np.stack([oracle.canny(f, 1.0, lo, hi) for f in frames])
                d_in, d_out = c.malloc(frames.nbytes), c.malloc(frames.nbytes * 2)
                try:
                    c.h2d(d_in, frames)
                    c.dev_canny(d_in, 1.0, lo, hi, h, w, n, d_out)
                    got = np.empty(frames.shape, np.int16)
                    c.d2h(got, d_out)
                finally:
                    c.free(d_in)
                    c.free(d_out)
                bad = np.argwhere(got != want)
                assert bad.size == 0, (fuse, lo, hi, h, w, n, bad[:5].tolist())


@pytest.mark.parametrize("gpath,spath", [(1, 1), (2, 2)])
def test_pipeline_paths_batched(hip, gpath, spath):
    frames = np.stack([_mixed(270, 520, 50 + i) for i in range(5)])
    want = np.stack([oracle.canny(f, 1.4, 50, 150) for f in frames])
    with hip.Context(0) as c:
        c.set_option("gaussian_path", gpath)
        c.set_option("sobel_nms_path", spath)
        d_in, d_out = c.malloc(frames.nbytes), c.malloc(frames.nbytes * 2)
        try:
            c.h2d(d_in, frames)
            c.dev_canny(d_in, 1.4, 50, 150, 270, 520, 5, d_out)
            got = np.empty(frames.shape, np.int16)
            c.d2h(got, d_out)
        finally:
            c.free(d_in)
            c.free(d_out)
        assert np.array_equal(got, want)


def test_canny_unaligned_device_buffers(hip):
    """Input and output planes that are only 1- / 2-byte aligned (the kernels use 4-, 8- and 16-byte accesses)."""
    h, w, n = 70, 520, 3
    frames = np.stack([_mixed(h, w, 90 + i) for i in range(n)])
    want = np.stack([oracle.canny(f, 1.4, 50, 150) for f in frames])
    with hip.Context(0) as c:
        d_in, d_out = c.malloc(frames.nbytes + 64), c.malloc(frames.nbytes * 2 + 64)
        try:
            for in_off, out_off in ((1, 2), (3, 6), (5, 14)):
                c.h2d(d_in + in_off, frames)
                c.dev_canny(d_in + in_off, 1.4, 50, 150, h, w, n, d_out + out_off)
                got = np.empty(frames.shape, np.int16)
                c.d2h(got, d_out + out_off)
                assert np.array_equal(got, want), (in_off, out_off)
        finally:
            c.free(d_in)
            c.free(d_out)


def _serpentine_image(h, w):
    """A dim one-pixel-wide serpentine line with one bright end: after Sobel+NMS its flanks are weak edges that
    hysteresis can only reach by walking the whole line from the bright (strong) end, tile after tile."""
    img = np.zeros((h, w), np.uint8)
    rows = list(range(4, h - 4, 8))
    for k, r in enumerate(rows):
        img[r, 4:w - 4] = 60
        if k + 1 < len(rows):
            c = w - 5 if k % 2 == 0 else 4
            img[r:rows[k + 1] + 1, c] = 60
    img[rows[0], 4:12] = 255
    return img


@pytest.mark.parametrize("overlap", [0, 1])
def test_canny_overlapped_halves(hip, overlap):
    """canny() on >= 16 frames runs the two halves' propagations on two streams; frames 2 and 11 need far more
    than one chunk of sweeps, so both lanes go through the relaunch loop."""
    h, w, n = 200, 328, 16
    frames = np.stack([_mixed(h, w, 70 + i) for i in range(n)])
    frames[2] = _serpentine_image(h, w)
    frames[11] = _serpentine_image(h, w)[::-1].copy()
    want = np.stack([oracle.canny(f, 0.3, 50, 250) for f in frames])
    assert np.count_nonzero(want[2]) > 2000  # the walk really happens
    with hip.Context(0) as c:
        c.set_option("overlap_hysteresis", overlap)
        d_in, d_out = c.malloc(frames.nbytes), c.malloc(frames.nbytes * 2)
        try:
            c.h2d(d_in, frames)
            for _ in range(2):  # twice: the second call reuses every workspace and event
                c.dev_canny(d_in, 0.3, 50, 250, h, w, n, d_out)
                got = np.empty(frames.shape, np.int16)
                c.d2h(got, d_out)
                bad = np.argwhere(got != want)
                assert bad.size == 0, (overlap, bad[:5].tolist())
            assert c.last_hysteresis_iterations > 8
        finally:
            c.free(d_in)
            c.free(d_out)


@pytest.mark.parametrize("overlap", [0, 1])
def test_canny_stream_of_batches(hip, overlap):
    """canny_hip_dev_canny_stream leaves each batch's sweeps in flight (queued before the next batch's Gaussian,
    or with stream_overlap=1 on a second stream beside it).  Batches
    of different content (one of them needs several chunks of sweeps), separate and reused output buffers, a
    plain call in between, a shape the fused kernel does not take, and flush / synchronize as the last word."""
    h, w, n = 200, 328, 6
    batches = []
    for b in range(4):
        frames = np.stack([_mixed(h, w, 300 + 10 * b + i) for i in range(n)])
        if b == 1:
            frames[3] = _serpentine_image(h, w)
        batches.append(frames)
    want = [np.stack([oracle.canny(f, 0.3, 50, 250) for f in fr]) for fr in batches]
    odd = np.stack([_mixed(h, w - 3, 900 + i) for i in range(2)])
    want_odd = np.stack([oracle.canny(f, 0.3, 50, 250) for f in odd])
    with hip.Context(0) as c:
        c.set_option("stream_overlap", overlap)
        nbytes = batches[0].nbytes
        d_in = [c.malloc(nbytes) for _ in range(2)]
        d_out = [c.malloc(nbytes * 2) for _ in range(4)]
        d_odd_in, d_odd_out = c.malloc(odd.nbytes), c.malloc(odd.nbytes * 2)
        try:
            def fetch(ptr, shape=batches[0].shape):
                got = np.empty(shape, np.int16)
                c.d2h(got, ptr)
                return got

            # 1) four batches back to back into four buffers (inputs double-buffered), then flush
            for b in range(4):
                c.h2d(d_in[b % 2], batches[b])
                c.dev_canny_stream(d_in[b % 2], 0.3, 50, 250, h, w, n, d_out[b])
                if b:  # the previous batch is complete as soon as this call has returned
                    assert np.array_equal(fetch(d_out[b - 1]), want[b - 1]), b - 1
            c.dev_canny_stream_flush()
            assert np.array_equal(fetch(d_out[3]), want[3])
            c.dev_canny_stream_flush()  # nothing pending: no-op

            # 2) one output buffer reused by every call; synchronize() finishes the last one
            for rnd in range(2):
                for b in (1, 2, 0):
                    c.h2d(d_in[0], batches[b])
                    c.dev_canny_stream(d_in[0], 0.3, 50, 250, h, w, n, d_out[0])
                c.synchronize()
                assert np.array_equal(fetch(d_out[0]), want[0]), rnd

            # 3) a plain call (and the stage API) while a streamed batch is in flight
            c.h2d(d_in[0], batches[1])
            c.h2d(d_in[1], batches[2])
            c.dev_canny_stream(d_in[0], 0.3, 50, 250, h, w, n, d_out[1])
            c.dev_canny(d_in[1], 0.3, 50, 250, h, w, n, d_out[2])
            assert np.array_equal(fetch(d_out[1]), want[1])
            assert np.array_equal(fetch(d_out[2]), want[2])

            # 4) a width the fused kernel does not take runs as a plain call, with a streamed batch before it
            c.dev_canny_stream(d_in[0], 0.3, 50, 250, h, w, n, d_out[3])
            c.h2d(d_odd_in, odd)
            c.dev_canny_stream(d_odd_in, 0.3, 50, 250, h, w - 3, 2, d_odd_out)
            assert np.array_equal(fetch(d_odd_out, odd.shape), want_odd)
            assert np.array_equal(fetch(d_out[3]), want[1])
        finally:
            for p in d_in + d_out + [d_odd_in, d_odd_out]:
                c.free(p)


@pytest.mark.parametrize("seed", range(6))
def test_canny_random_shapes_and_parameters(hip, seed):
    """Seeded fuzz of the whole pipeline: shapes around the kernels' strip (496/248/240 columns), segment (32/64
    rows) and tile (64 x 64) boundaries, widths with and without the % 8 the fused kernel wants, random sigma,
    thresholds and batch size -- every frame bit-identical to the oracle, through both canny entry points."""
    rng = np.random.default_rng(1000 + seed)
    edges_w = [8, 56, 64, 72, 240, 248, 256, 488, 496, 504, 512, 736, 744, 992, 1000]
    edges_h = [2, 3, 31, 32, 33, 63, 64, 65, 66, 127, 128, 130, 191, 193]
    with hip.Context(0) as c:
        for _ in range(7):
            w = int(rng.choice(edges_w)) + int(rng.integers(-3, 4)) * int(rng.integers(0, 2))
            h = int(rng.choice(edges_h)) + int(rng.integers(0, 3))
            w, h = max(2, w), max(2, h)
            n = int(rng.integers(1, 4))
            sigma = float(rng.choice([0.4, 0.8, 1.0, 1.4, 1.7, 2.0, 2.4]))
            lo = int(rng.integers(1, 120))
            hi = int(rng.integers(lo + 1, 256))
            gens = (_noise, _mixed)
            frames = np.stack([gens[int(rng.integers(0, 2))](h, w, int(rng.integers(0, 1 << 30))) for _ in range(n)])
            want = np.stack([oracle.canny(f, sigma, lo, hi) for f in frames])
            d_in, d_out = c.malloc(frames.nbytes), c.malloc(frames.nbytes * 2)
            try:
                c.h2d(d_in, frames)
                got = np.empty(frames.shape, np.int16)
                c.dev_canny(d_in, sigma, lo, hi, h, w, n, d_out)
                c.d2h(got, d_out)
                assert np.array_equal(got, want), ("dev_canny", h, w, n, sigma, lo, hi)
                c.dev_canny_stream(d_in, sigma, lo, hi, h, w, n, d_out)
                c.dev_canny_stream_flush()
                c.d2h(got, d_out)
                assert np.array_equal(got, want), ("dev_canny_stream", h, w, n, sigma, lo, hi)
            finally:
                c.free(d_in)
                c.free(d_out)


@pytest.mark.parametrize("seg", [8, 12, 20, 33, 100])
def test_canny_with_any_sobel_segment_length(hip, seg, sobel_px):
    """tune_sobel_seg is free-form; the fused kernel's staged plane bytes leave in groups of 8 rows, so the launcher
    rounds what it is given -- whatever the knob says, the result is the oracle's."""
    h, w, n = 150, 1000, 2
    frames = np.stack([_mixed(h, w, 40 + i) for i in range(n)])
    want = np.stack([oracle.canny(f, 1.0, 40, 120) for f in frames])
    with hip.Context(0) as c:
        c.set_option("tune_sobel_seg", seg)
        d_in, d_out = c.malloc(frames.nbytes), c.malloc(frames.nbytes * 2)
        try:
            c.h2d(d_in, frames)
            c.dev_canny(d_in, 1.0, 40, 120, h, w, n, d_out)
            got = np.empty(frames.shape, np.int16)
            c.d2h(got, d_out)
            assert np.array_equal(got, want)
        finally:
            c.free(d_in)
            c.free(d_out)


def test_canny_large_resident_batch_of_4k_frames(hip):
    """BASELINE config 5's per-GPU share in shape (hundreds of resident 4K frames in one call): 512 frames = 4.2 Gpx,
    i.e. 34 GB of planes and workspaces -- 32-bit pixel counts, tile counts and launch grids all go far past what
    the other tests reach.  Eight distinct frames cycle through the batch; the first and the last copy of each are
    compared with the oracle, and a checksum of checksums over ALL frames checks that equal inputs gave equal maps."""
    h, w, n, distinct = 2160, 3840, 512, 8
    base = [synth_frame(h, w, 4200 + i) for i in range(distinct)]
    want = [oracle.canny(f, 1.4, 50, 150) for f in base]
    frame_px = h * w
    with hip.Context(0) as c:
        d_in, d_out = c.malloc(n * frame_px), c.malloc(n * frame_px * 2)
        try:
            for i in range(n):
                c.h2d(d_in + i * frame_px, base[i % distinct])
            c.dev_canny(d_in, 1.4, 50, 150, h, w, n, d_out)
            got = np.empty((h, w), np.int16)
            sums = []
            for i in range(n):
                if i < distinct or i >= n - distinct:
                    c.d2h(got, d_out + i * frame_px * 2)
                    assert np.array_equal(got, want[i % distinct]), i
            # every frame's edge count, taken on the device side of the copy: one u8 narrowing + host sum per frame
            # would move 4 GB; instead fetch one row band per frame that crosses tile and segment boundaries
            band = np.empty((80, w), np.int16)
            for i in range(n):
                c.d2h(band, d_out + (i * frame_px + 1000 * w) * 2)
                sums.append(int(np.count_nonzero(band)))
            for i in range(n):
                assert sums[i] == sums[i % distinct], i
                assert sums[i] == int(np.count_nonzero(want[i % distinct][1000:1080])), i
        finally:
            c.free(d_in)
            c.free(d_out)


# ---------------------------------------------------------------------------------------------
# Round 2: the per-frame tail kernel of the propagation, and the min_val > 255 domain
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tail", [1, 0], ids=["tail_kernel", "multi_launch"])
def test_canny_hysteresis_tail_kernel(hip, tail):
    """canny() finishes the propagation with ONE launch (a workgroup per frame loops over that frame's queue until it
    stays empty).  Frames of very different depth in one batch -- serpentines that need hundreds of sweeps next to
    frames that need two -- several shapes (one tile, one tile row, many tiles), and repeated calls on the same
    context; `hysteresis_tail` = 0 is the multi-launch scheme with the host poll.  Both against the oracle."""
    with hip.Context(0) as c:
        c.set_option("hysteresis_tail", tail)
        for h, w, n in ((200, 328, 9), (64, 64, 3), (40, 1000, 4), (2, 8, 2), (330, 136, 5)):
            frames = np.stack([_mixed(h, w, 500 + i) for i in range(n)])
            if h >= 40:
                frames[1] = _serpentine_image(h, w)
                frames[n - 1] = _serpentine_image(h, w)[::-1, ::-1].copy()
            want = np.stack([oracle.canny(f, 0.3, 50, 250) for f in frames])
            d_in, d_out = c.malloc(frames.nbytes), c.malloc(frames.nbytes * 2)
            try:
                c.h2d(d_in, frames)
                for rep in range(2):
                    c.dev_canny(d_in, 0.3, 50, 250, h, w, n, d_out)
                    got = np.empty(frames.shape, np.int16)
                    c.d2h(got, d_out)
                    bad = np.argwhere(got != want)
                    assert bad.size == 0, (tail, h, w, n, rep, bad[:5].tolist())
                if h == 200:
                    assert c.last_hysteresis_iterations > 8  # the serpentine really needs a long walk
            finally:
                c.free(d_in)
                c.free(d_out)


def test_canny_frames_beyond_the_tail_kernels_tile_limit(hip):
    """More than 4096 tiles per frame: the propagation goes back to the multi-launch scheme (one workgroup per
    frame would be too few).  64 rows x 262,208 columns = 1 x 4097 tiles."""
    h, w = 64, 64 * 4097
    img = _mixed(h, w, 77)
    with hip.Context(0) as c:
        assert np.array_equal(c.canny(img, 1.0, 50, 150), oracle.canny(img, 1.0, 50, 150))


def test_thresholds_above_edge_value_are_a_domain_error(ctx, hip):
    """min_val > 255 >= max_val: the reference overwrites reached pixels with EDGE = 255 while its scan is still
    running, so a reached pixel the scan has not passed yet fails `< minVal` and is zeroed again
    (src/utils.cpp:327-334): the result depends on the scan order."""
    a = np.array([[300, 300, 0, 0], [0, 0, 0, 0]], np.int16)
    assert oracle.hysteresis(a, 300, 100).ravel().tolist() == [255, 0, 0, 0, 0, 0, 0, 0]  # what the reference does
    with pytest.raises(hip.CannyHipError) as ei:
        ctx.hysteresis(a, 300, 100)
    assert ei.value.status == 5
    with pytest.raises(hip.CannyHipError) as ei:
        ctx.canny(_mixed(16, 16, 1), 1.0, 300, 100)
    assert ei.value.status == 5
    with pytest.raises(hip.CannyHipError) as ei:
        ctx.find_edge_pixels(a, np.zeros(a.shape, np.uint8), 0, 300, 400)
    assert ei.value.status == 5
    # both thresholds above 255: everything ends as 0 whatever the order -- accepted, and equal to the oracle
    assert np.array_equal(ctx.hysteresis(a, 300, 400), oracle.hysteresis(a, 300, 400))
    img = _mixed(24, 40, 3)
    assert np.array_equal(ctx.canny(img, 1.0, 300, 400), oracle.canny(img, 1.0, 300, 400))
