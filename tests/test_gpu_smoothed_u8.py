"""The u8 smoothed plane: `(short)(sum/count)` of the reference's Gaussian always lies in
[0,255] (src/utils.cpp:62), so canny() may hand it from the Gaussian to the fused Sobel+NMS kernel as bytes.
Everything the option touches is compared with the oracle bit for bit: the byte-storing Gaussian,
the byte-reading Sobel+NMS, and canny() with "smoothed_u8" = 1 on shapes with and without the
fused path, batches, the stream API and unaligned device buffers."""
import numpy as np
import pytest

import oracle
from canny_edge_amd.synth import synth_frame

pytestmark = pytest.mark.gpu


def _noise(h, w, seed):
    return np.random.default_rng(seed).integers(0, 256, size=(h, w), dtype=np.uint8)


def _mixed(h, w, seed):
    img = synth_frame(h, w, seed).astype(np.int16)
    yy, xx = np.mgrid[0:h, 0:w]
    img = img // 2 + ((xx * 3 + yy * 2) % 128).astype(np.int16)
    return np.clip(img, 0, 255).astype(np.uint8)


SHAPES = [(2, 8), (3, 5), (17, 19), (64, 64), (65, 72), (97, 131), (130, 496), (100, 504), (63, 520), (129, 1000),
          (200, 1488), (70, 2048), (300, 250)]


@pytest.mark.parametrize("sigma", [0.5, 1.0, 1.4, 2.0, 2.6])
def test_gaussian_u8_plane_equals_oracle(hip, sigma):
    with hip.Context(0) as c:
        for h, w in SHAPES + [(1, 1), (1, 40), (40, 1), (6, 3)]:
            frames = np.stack([_mixed(h, w, 7), _noise(h, w, 8), np.full((h, w), 255, np.uint8)])
            want = np.stack([oracle.gaussian(f, sigma) for f in frames])
            assert want.min() >= 0 and want.max() <= 255
            d_in, d_out = c.malloc(frames.nbytes), c.malloc(frames.nbytes)
            try:
                c.h2d(d_in, frames)
                if w < 4:  # the marching kernels need whole dwords inside a row; canny() keeps the s16 plane there
                    with pytest.raises(hip.CannyHipError):
                        c.dev_gaussian_u8(d_in, sigma, h, w, len(frames), d_out)
                    if h >= 2 and w >= 2:
                        got = c.canny(frames[0], sigma, 50, 150)
                        assert c.get_option("last_canny_smoothed_u8") == 0
                        assert np.array_equal(got, oracle.canny(frames[0], sigma, 50, 150)), (sigma, h, w)
                    continue
                c.dev_gaussian_u8(d_in, sigma, h, w, len(frames), d_out)
                got = np.empty(frames.shape, np.uint8)
                c.d2h(got, d_out)
            finally:
                c.free(d_in)
                c.free(d_out)
            assert np.array_equal(got.astype(np.int16), want), (sigma, h, w)


def test_sobel_nms_from_u8_plane_equals_oracle(hip):
    with hip.Context(0) as c:
        for h, w in [s for s in SHAPES if s[0] >= 2 and s[1] >= 2] + [(2, 2), (9, 2), (2, 9)]:
            sm = np.stack([oracle.gaussian(_mixed(h, w, 3), 1.0), _noise(h, w, 4).astype(np.int16)])
            want = []
            for plane in sm:
                mag, ang = oracle.sobel(plane)
                want.append(oracle.nms(mag, ang))
            want = np.stack(want)
            sm8 = sm.astype(np.uint8)
            d_in, d_out = c.malloc(sm8.nbytes), c.malloc(sm8.nbytes * 2)
            try:
                c.h2d(d_in, sm8)
                c.dev_sobel_nms_u8in(d_in, h, w, len(sm8), d_out)
                got = np.empty(sm.shape, np.int16)
                c.d2h(got, d_out)
            finally:
                c.free(d_in)
                c.free(d_out)
            assert np.array_equal(got, want), (h, w)


@pytest.mark.parametrize("mode", [1, 0], ids=["u8_plane_default", "s16_plane"])
@pytest.mark.parametrize("lo,hi", [(50, 150), (1, 1), (100, 50), (0, 100), (255, 256)])
def test_canny_with_u8_smoothed_plane(hip, mode, lo, hi):
    """Shapes with width % 8 != 0 and min_val = 0 do not take the fused path: the option must then be ignored."""
    with hip.Context(0) as c:
        c.set_option("smoothed_u8", mode)
        for h, w in SHAPES:
            for sigma in (1.0, 1.4):
                n = 3
                frames = np.stack([_mixed(h, w, 10 * i + 1) if i != 1 else _noise(h, w, 2) for i in range(n)])
                want = np.stack([oracle.canny(f, sigma, lo, hi) for f in frames])
                d_in, d_out = c.malloc(frames.nbytes + 16), c.malloc(frames.nbytes * 2 + 16)
                try:
                    c.h2d(d_in + 1, frames)  # unaligned on purpose
                    c.dev_canny(d_in + 1, sigma, lo, hi, h, w, n, d_out + 2)
                    got = np.empty(frames.shape, np.int16)
                    c.d2h(got, d_out + 2)
                finally:
                    c.free(d_in)
                    c.free(d_out)
                assert np.array_equal(got, want), (mode, lo, hi, h, w, sigma)


@pytest.mark.parametrize("mode", [1, 0], ids=["u8_plane_default", "s16_plane"])
def test_canny_u8_smoothed_plane_large_windows_fall_back(hip, mode):
    """sigma 3.0 -> window 19: no marching Gaussian, so no byte plane; the result must not change."""
    img = _mixed(120, 200, 5)
    with hip.Context(0) as c:
        c.set_option("smoothed_u8", mode)
        for sigma in (3.0, 0.1, 2.6):
            assert np.array_equal(c.canny(img, sigma, 30, 90), oracle.canny(img, sigma, 30, 90)), sigma


@pytest.mark.parametrize("mode", [1, 0], ids=["u8_plane_default", "s16_plane"])
def test_stream_of_batches_and_host_batch_with_u8_smoothed_plane(hip, mode):
    h, w, n = 270, 480, 6
    batches = [np.stack([synth_frame(h, w, 100 * b + i) for i in range(n)]) for b in range(4)]
    want = [np.stack([oracle.canny(f, 1.4, 50, 150) for f in fr]) for fr in batches]
    with hip.Context(0) as c:
        c.set_option("smoothed_u8", mode)
        d_in = [c.malloc(batches[0].nbytes) for _ in range(2)]
        d_out = [c.malloc(batches[0].nbytes * 2) for _ in range(2)]
        try:
            got = []
            for b, fr in enumerate(batches):
                c.h2d(d_in[b % 2], fr)
                c.dev_canny_stream(d_in[b % 2], 1.4, 50, 150, h, w, n, d_out[b % 2])
                if b > 0:
                    out = np.empty(fr.shape, np.int16)
                    c.d2h(out, d_out[(b - 1) % 2])
                    got.append(out)
            c.dev_canny_stream_flush()
            out = np.empty(batches[-1].shape, np.int16)
            c.d2h(out, d_out[(len(batches) - 1) % 2])
            got.append(out)
        finally:
            for p in d_in + d_out:
                c.free(p)
        for b in range(len(batches)):
            assert np.array_equal(got[b], want[b]), b
        # 4K-wide frame through the host API (many strips per row)
        big = synth_frame(96, 3840, 9)
        assert np.array_equal(c.canny(big, 1.4, 50, 150), oracle.canny(big, 1.4, 50, 150))


def test_u8_plane_is_the_default_and_reported(hip):
    """Round 3: canny() keeps the smoothed plane as bytes by default; "last_canny_smoothed_u8" tells which plane the last
    call really used (bench.py prices the Sobel+NMS kernel it timed with the bytes that kernel moved)."""
    img = _mixed(128, 200, 3)        # width % 8 == 0: fused path, marching Gaussian at sigma 1.4
    odd = _mixed(128, 203, 4)        # width % 8 != 0: the fused kernel does not take it
    with hip.Context(0) as c:
        assert c.get_option("smoothed_u8") == 1
        assert np.array_equal(c.canny(img, 1.4, 50, 150), oracle.canny(img, 1.4, 50, 150))
        assert c.get_option("last_canny_smoothed_u8") == 1
        assert np.array_equal(c.canny(img, 3.0, 50, 150), oracle.canny(img, 3.0, 50, 150))   # window 19
        assert c.get_option("last_canny_smoothed_u8") == 0
        assert np.array_equal(c.canny(odd, 1.4, 50, 150), oracle.canny(odd, 1.4, 50, 150))
        assert c.get_option("last_canny_smoothed_u8") == 0
        c.set_option("smoothed_u8", 0)
        assert np.array_equal(c.canny(img, 1.4, 50, 150), oracle.canny(img, 1.4, 50, 150))
        assert c.get_option("last_canny_smoothed_u8") == 0
        with pytest.raises(hip.CannyHipError):
            c.get_option("no_such_option")
