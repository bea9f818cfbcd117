"""The host batch pipeline (BASELINE configs 3 and 5) against the oracle, frame by frame.

`canny_hip_canny_batch` / `_u8` cut a batch into chunks that move through H2D -> kernels -> D2H on separate
streams; `canny_hip_canny_multi_gpu` shards a batch over devices and runs that pipeline per shard.  The reference
has no batch entry point: it calls canny() once per frame (src/main.cpp:120-137, src/utils.cpp:429-478), so the
contract is "output frame i == canny(frame i)" and that is what every test here checks, with the chunk size,
the number of pipelines and the kind of host memory forced through every combination:

  * chunk boundaries inside the batch, a short last chunk, more pipelines than chunks;
  * pageable -> pageable, pinned -> pinned and both mixed cases;
  * s16 and u8 edge maps;
  * the context's cached staging reused by a second call with another shape in between;
  * the sharder with more shards than devices (shards wrap around the devices of the box).
"""
import numpy as np
import pytest

import oracle
from canny_edge_amd.synth import synth_frame

pytestmark = pytest.mark.gpu

SIGMA, LO, HI = 1.0, 50, 150


def _frames(n, h, w, seed0=0, distinct=None):
    distinct = n if distinct is None else distinct
    base = [synth_frame(h, w, seed0 + i) for i in range(distinct)]
    return np.stack([base[i % distinct] for i in range(n)])


_ORACLE_CACHE = {}


def _want(frames, sigma=SIGMA, lo=LO, hi=HI):
    """Oracle edge maps, one canny() per frame like the reference's capture loop."""
    out = np.empty(frames.shape, np.int16)
    for i, f in enumerate(frames):
        key = (f.tobytes(), f.shape, sigma, lo, hi)
        if key not in _ORACLE_CACHE:
            _ORACLE_CACHE[key] = oracle.canny(f, sigma, lo, hi)
        out[i] = _ORACLE_CACHE[key]
    return out


def _assert_frames_equal(got, want, what):
    assert got.shape == want.shape
    for i in range(len(want)):
        if not np.array_equal(got[i], want[i]):
            bad = int((got[i] != want[i]).sum())
            raise AssertionError(f"{what}: frame {i} of {len(want)} differs from oracle.canny in {bad} pixels")


@pytest.fixture(scope="module")
def batch_270(hip):
    f = _frames(43, 270, 480, seed0=100)
    return f, _want(f)


@pytest.fixture(scope="module")
def batch_odd(hip):
    f = _frames(41, 37, 53, seed0=200)  # odd width: the non-fused kernels, and 1961-byte frames
    return f, _want(f)


def _run(ctx, frames, u8, mem, out_dtype):
    n, h, w = frames.shape
    pin_in = mem in ("pinned", "pinned_in")
    pin_out = mem in ("pinned", "pinned_out")
    src = ctx.pinned_array(frames.shape, np.uint8) if pin_in else np.empty(frames.shape, np.uint8)
    src[...] = frames
    dst = ctx.pinned_array(frames.shape, out_dtype) if pin_out else np.empty(frames.shape, out_dtype)
    dst[...] = 0x55 if u8 else 0x5555  # stale bytes must not survive
    got = ctx.canny_batch(src, SIGMA, LO, HI, out=dst, u8=u8)
    return np.array(got)  # copy out of pinned memory before the context frees it


@pytest.mark.parametrize("u8", [False, True], ids=["s16", "u8"])
@pytest.mark.parametrize("mem", ["pageable", "pinned", "pinned_in", "pinned_out"])
@pytest.mark.parametrize("workers", [1, 2, 3, 4])
@pytest.mark.parametrize("pipe_mode", [0, 1, 2], ids=["auto", "three_streams", "one_stream"])
def test_batch_across_chunk_and_pipeline_boundaries(hip, batch_270, batch_odd, pipe_mode, workers, mem, u8):
    """43 x 270x480 in 1 MB chunks = 8 frames per chunk, 6 chunks, the last one 3 frames; then 41 x 37x53 in
    chunks of 6 frames (7 chunks, last one 5) on the SAME context, then the first shape again: the staging
    buffers and pipelines cached in the context are reused across calls with different shapes."""
    dtype = np.uint8 if u8 else np.int16
    with hip.Context(0) as ctx:
        ctx.set_option("tune_batch_workers", workers)
        ctx.set_option("tune_batch_pipe_mode", pipe_mode)
        for frames, want, chunk_opt in ((batch_270[0], batch_270[1], ("tune_batch_chunk_mb", 1)),
                                        (batch_odd[0], batch_odd[1], ("tune_batch_chunk_frames", 6)),
                                        (batch_270[0][:19], batch_270[1][:19], ("tune_batch_chunk_frames", 4))):
            ctx.set_option("tune_batch_chunk_mb", 0)
            ctx.set_option("tune_batch_chunk_frames", 0)
            ctx.set_option(*chunk_opt)
            got = _run(ctx, frames, u8, mem, dtype)
            assert got.dtype == dtype
            _assert_frames_equal(got.astype(np.int16), want, f"mode={pipe_mode} workers={workers} {mem} u8={u8} {chunk_opt}")


def test_batch_more_pipelines_than_chunks_and_single_frame_chunks(hip, batch_270):
    frames, want = batch_270[0][:5], batch_270[1][:5]
    with hip.Context(0) as ctx:
        ctx.set_option("tune_batch_workers", 4)
        ctx.set_option("tune_batch_chunk_frames", 3)  # 2 chunks for 4 pipelines
        _assert_frames_equal(ctx.canny_batch(frames, SIGMA, LO, HI), want, "2 chunks / 4 pipelines")
        ctx.set_option("tune_batch_chunk_frames", 1)  # one frame per chunk: 5 chunks
        _assert_frames_equal(ctx.canny_batch(frames, SIGMA, LO, HI), want, "1-frame chunks")
        got8 = ctx.canny_batch(frames[:1], SIGMA, LO, HI, u8=True)
        _assert_frames_equal(got8.astype(np.int16), want[:1], "single frame, u8")


def test_batch_default_tuning_and_other_parameters(hip, batch_270):
    """Automatic chunking (no options set), another sigma and thresholds, frames that differ in every chunk."""
    frames = batch_270[0]
    with hip.Context(0) as ctx:
        for sigma, lo, hi in ((1.4, 30, 90), (0.5, 100, 200), (2.0, 20, 40)):
            want = _want(frames[:12], sigma, lo, hi)
            _assert_frames_equal(ctx.canny_batch(frames[:12], sigma, lo, hi), want, f"sigma={sigma}")
            _assert_frames_equal(ctx.canny_batch(frames[:12], sigma, lo, hi, u8=True).astype(np.int16), want,
                                 f"sigma={sigma} u8")


def test_batch_input_is_not_modified_and_output_is_fully_written(hip, batch_270):
    frames, want = batch_270[0][:17], batch_270[1][:17]
    src = frames.copy()
    with hip.Context(0) as ctx:
        ctx.set_option("tune_batch_chunk_frames", 5)
        out = np.full(frames.shape, -1, np.int16)
        ctx.canny_batch(src, SIGMA, LO, HI, out=out)
    assert np.array_equal(src, frames)
    assert set(np.unique(out)) <= {0, 255}
    _assert_frames_equal(out, want, "prefilled output")


@pytest.mark.parametrize("shards", [1, 2, 3, 5])
@pytest.mark.parametrize("u8", [False, True], ids=["s16", "u8"])
def test_multi_gpu_sharder_against_oracle(hip, batch_270, shards, u8):
    """canny_hip_canny_multi_gpu with `shards` shards.  A one-GPU box runs them on device (shard % devices) when
    "allow_device_reuse" is set, so the shard arithmetic (contiguous ranges, uneven split, one context and one
    pipeline per shard, cached between calls) is exercised with N > 1 exactly as on an 8-GPU node."""
    frames, want = batch_270
    hip.multi_gpu_set_option("allow_device_reuse", 1)
    hip.multi_gpu_set_option("tune_batch_chunk_frames", 4)
    try:
        for n in (43, 7, 2):  # 7 frames over 5 shards: shards of 2,2,1,1,1; 2 frames over 5 shards: empty shards
            got = hip.canny_multi_gpu(frames[:n], SIGMA, LO, HI, n_devices=shards, u8=u8)
            assert got.dtype == (np.uint8 if u8 else np.int16)
            _assert_frames_equal(got.astype(np.int16), want[:n], f"shards={shards} n={n} u8={u8}")
    finally:
        hip.multi_gpu_set_option("tune_batch_chunk_frames", 0)
        hip.multi_gpu_set_option("allow_device_reuse", 0)
        hip.multi_gpu_release()


def test_multi_gpu_pinned_buffers_and_repeat_calls(hip, batch_270):
    frames, want = batch_270[0][:24], batch_270[1][:24]
    hip.multi_gpu_set_option("allow_device_reuse", 1)
    try:
        with hip.Context(0) as ctx:
            src = ctx.pinned_array(frames.shape, np.uint8)
            src[...] = frames
            dst = ctx.pinned_array(frames.shape, np.int16)
            for _ in range(3):  # cached per-device contexts are reused
                dst[...] = -1
                hip.canny_multi_gpu(src, SIGMA, LO, HI, n_devices=2, out=dst)
                _assert_frames_equal(np.array(dst), want, "pinned multi-gpu")
    finally:
        hip.multi_gpu_set_option("allow_device_reuse", 0)
        hip.multi_gpu_release()


def test_batch_on_registered_caller_memory(hip, batch_270):
    """Memory the caller allocated itself (the reference's frames are new[] arrays), page-locked with
    canny_hip_host_register: the pinned path of the pipeline, then unregistered: the staging path again."""
    frames, want = batch_270
    src = frames.copy()
    dst = np.full(frames.shape, -1, np.int16)
    with hip.Context(0) as ctx:
        ctx.set_option("tune_batch_chunk_frames", 5)
        ctx.host_register(src)
        ctx.host_register(dst)
        try:
            ctx.canny_batch(src, SIGMA, LO, HI, out=dst)
            _assert_frames_equal(dst, want, "registered buffers")
            dst[...] = -1
            ctx.canny_batch(src, SIGMA, LO, HI, out=dst)
            _assert_frames_equal(dst, want, "registered buffers, second call")
        finally:
            ctx.host_unregister(src)
            ctx.host_unregister(dst)
        dst[...] = -1
        ctx.canny_batch(src, SIGMA, LO, HI, out=dst)
        _assert_frames_equal(dst, want, "after unregistering")
        with pytest.raises(hip.CannyHipError):
            ctx.host_unregister(src)  # not registered any more


def test_batch_true_size_1024_frames_of_1080p(hip):
    """BASELINE config 3 at its real size: 1024 x 1920x1080, sigma 1.0, 16 distinct frames cycled, every one of the
    1024 output frames compared with its oracle map (pinned buffers, default tuning, s16 then u8)."""
    n, h, w = 1024, 1080, 1920
    base = np.stack([synth_frame(h, w, 42 + i) for i in range(16)])
    want = np.stack([oracle.canny(f, 1.0, 50, 150) for f in base])
    assert all(int(np.count_nonzero(m)) > 1000 for m in want)  # every frame has real work for hysteresis
    with hip.Context(0) as ctx:
        src = ctx.pinned_array((n, h, w), np.uint8)
        for i in range(n):
            src[i] = base[i % 16]
        dst = ctx.pinned_array((n, h, w), np.int16)
        dst[...] = -1
        ctx.canny_batch(src, 1.0, 50, 150, out=dst)
        for i in range(n):
            assert np.array_equal(dst[i], want[i % 16]), f"s16 frame {i}"
        dst8 = ctx.pinned_array((n, h, w), np.uint8)
        dst8[...] = 7
        ctx.canny_batch(src, 1.0, 50, 150, out=dst8, u8=True)
        want8 = want.astype(np.uint8)
        for i in range(n):
            assert np.array_equal(dst8[i], want8[i % 16]), f"u8 frame {i}"
        # pageable in and out, a few chunks only (the staging path at full frame size)
        got = ctx.canny_batch(np.array(src[:40]), 1.0, 50, 150)
        for i in range(40):
            assert np.array_equal(got[i], want[i % 16]), f"pageable frame {i}"


def test_batch_chunk_never_exceeds_the_kernels_frame_limit(hip):
    """Tiny frames, many of them: a chunk must not exceed the 65535 frames the kernels take per launch
    (grid.z), whatever the megabyte budget says."""
    n, h, w = 70000, 4, 8
    rng = np.random.default_rng(5)
    frames = rng.integers(0, 256, size=(n, h, w), dtype=np.uint8)
    with hip.Context(0) as ctx:
        got = ctx.canny_batch(frames, 1.0, 20, 60)
    for i in list(range(0, n, 997)) + [65534, 65535, 65536, n - 1]:
        assert np.array_equal(got[i], oracle.canny(frames[i], 1.0, 20, 60)), f"frame {i}"
