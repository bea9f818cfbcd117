"""Edge maps as bit maps (canny_hip_canny_batch_bits / _multi_gpu_bits / canny_hip_dev_canny_bits): 1 bit per pixel,
rows MSB-first and padded to whole bytes.  The reference's map is a short plane of 0 / 255 (src/utils.h:5-6,
src/utils.cpp:478); unpacked, the bit map has to be exactly that plane -- frame by frame against oracle.canny -- and the
padding bits of every row have to be zero."""
import numpy as np
import pytest

import oracle
from canny_edge_amd.synth import synth_frame

pytestmark = pytest.mark.gpu

SIGMA, LO, HI = 1.0, 50, 150


def _frames(n, h, w, seed0):
    return np.stack([synth_frame(h, w, seed0 + i) for i in range(n)])


def _check(hip, bits, frames, want, what):
    n, h, w = frames.shape
    assert bits.dtype == np.uint8 and bits.shape == (n, h, (w + 7) // 8), what
    assert np.array_equal(bits, np.packbits(want != 0, axis=-1)), what      # includes the zero padding of every row
    assert np.array_equal(hip.unpack_bits(bits, w), want), what


@pytest.mark.parametrize("shape", [(270, 480), (37, 53), (64, 8), (9, 2), (2, 9), (120, 1001)])
@pytest.mark.parametrize("mem", ["pageable", "pinned"])
@pytest.mark.parametrize("pipe_mode", [0, 1, 2], ids=["auto", "three_streams", "one_stream"])
def test_bit_maps_across_chunk_boundaries(hip, shape, mem, pipe_mode):
    h, w = shape
    frames = _frames(23, h, w, 700 + h)
    want = np.stack([oracle.canny(f, SIGMA, LO, HI) for f in frames])
    with hip.Context(0) as ctx:
        ctx.set_option("tune_batch_pipe_mode", pipe_mode)
        ctx.set_option("tune_batch_workers", 2)
        for chunk in (5, 1, 0):  # 5 chunks with a short last one; one frame per chunk; automatic
            ctx.set_option("tune_batch_chunk_frames", chunk)
            if mem == "pinned":
                src = ctx.pinned_array(frames.shape, np.uint8)
                src[...] = frames
                dst = ctx.pinned_array(hip.bits_shape(frames.shape), np.uint8)
            else:
                src, dst = frames, np.empty(hip.bits_shape(frames.shape), np.uint8)
            dst[...] = 0xA5  # stale bytes must not survive
            got = np.array(ctx.canny_batch(src, SIGMA, LO, HI, out=dst, bits=True))
            _check(hip, got, frames, want, f"{shape} {mem} mode={pipe_mode} chunk={chunk}")
            # the three formats of one call agree
            assert np.array_equal(ctx.canny_batch(frames, SIGMA, LO, HI), want)
            assert np.array_equal(ctx.canny_batch(frames, SIGMA, LO, HI, u8=True), want.astype(np.uint8))


@pytest.mark.parametrize("shape", [(96, 256), (45, 77)])
def test_device_resident_bit_maps(hip, shape):
    h, w = shape
    frames = _frames(4, h, w, 900)
    want = np.stack([oracle.canny(f, 1.4, 40, 120) for f in frames])
    nbytes = 4 * h * ((w + 7) // 8)
    with hip.Context(0) as ctx:
        d_in, d_out = ctx.malloc(frames.nbytes), ctx.malloc(nbytes + 16)
        try:
            ctx.h2d(d_in, frames)
            for shift in (0, 1, 16):
                ctx.dev_canny_bits(d_in, 1.4, 40, 120, h, w, 4, d_out + shift)
                got = np.empty(hip.bits_shape(frames.shape), np.uint8)
                ctx.d2h(got, d_out + shift)
                _check(hip, got, frames, want, f"{shape} shift={shift}")
        finally:
            ctx.free(d_in)
            ctx.free(d_out)


@pytest.mark.parametrize("shards", [1, 3])
def test_sharded_bit_maps(hip, shards):
    frames = _frames(11, 120, 203, 40)
    want = np.stack([oracle.canny(f, SIGMA, LO, HI) for f in frames])
    hip.multi_gpu_set_option("allow_device_reuse", 1)
    hip.multi_gpu_set_option("tune_batch_chunk_frames", 2)
    try:
        got = hip.canny_multi_gpu(frames, SIGMA, LO, HI, n_devices=shards, bits=True)
        _check(hip, got, frames, want, f"{shards} shards")
    finally:
        hip.multi_gpu_set_option("allow_device_reuse", 0)
        hip.multi_gpu_set_option("tune_batch_chunk_frames", 0)
        hip.multi_gpu_release()


def test_bit_maps_full_size_1080p(hip):
    """64 x 1080p through the default pipeline: properties on every frame (a bit map is the s16 map, packed), oracle on
    a few of them."""
    base = _frames(8, 1080, 1920, 5)
    frames = np.stack([base[i % 8] for i in range(64)])
    with hip.Context(0) as ctx:
        bits = ctx.canny_batch(frames, SIGMA, LO, HI, bits=True)
        s16 = ctx.canny_batch(frames, SIGMA, LO, HI)
    assert np.array_equal(bits, np.packbits(s16 != 0, axis=-1))
    for i in (0, 7, 63):
        assert np.array_equal(hip.unpack_bits(bits[i:i + 1], 1920)[0], oracle.canny(frames[i], SIGMA, LO, HI))
