"""Concurrent callers: SURVEY.md 8(b) -- the reference's CPU functions are re-entrant (single-threaded, no globals), so
the replacement has to be callable from several host threads at once, each with a context of its own (one per GPU in
production; here they share the box's one GPU, which is the harder case for the runtime: every context has its own
streams, scratch planes, queues and pipelines on the same device).  ctypes drops the GIL for the duration of a call,
so these threads really are inside the library together."""
import threading

import numpy as np
import pytest

import oracle
from canny_edge_amd.synth import synth_frame

pytestmark = pytest.mark.gpu


def _worker(hip, tid, rounds, errors, barrier):
    try:
        shapes = [(97 + 16 * tid, 131 + 8 * tid), (240, 320 + 64 * tid), (1080, 1920) if tid % 2 else (64, 1024)]
        imgs = [synth_frame(h, w, 300 + 10 * tid + k) for k, (h, w) in enumerate(shapes)]
        sigma, lo, hi = (1.0, 1.4, 2.0, 0.5)[tid % 4], 40 + tid, 120 + 2 * tid
        want = [oracle.canny(im, sigma, lo, hi, stages=True) for im in imgs]
        batch = np.stack([synth_frame(120, 200, 500 + 7 * tid + k) for k in range(9)])
        want_batch = np.stack([oracle.canny(f, sigma, lo, hi) for f in batch])
        with hip.Context(0) as ctx:
            ctx.set_option("tune_batch_chunk_frames", 2 + tid % 3)   # several chunks per call
            barrier.wait(timeout=120)
            for r in range(rounds):
                for im, w in zip(imgs, want):
                    assert np.array_equal(ctx.canny(im, sigma, lo, hi), w["edges"]), (tid, r, im.shape, "canny")
                    sm = ctx.gaussian(im, sigma)
                    assert np.array_equal(sm, w["smoothed"]), (tid, r, im.shape, "gaussian")
                    mag, ang = ctx.sobel(sm)
                    nm = ctx.nms(mag, ang)
                    assert np.array_equal(nm, w["nms"]), (tid, r, im.shape, "nms")
                    assert np.array_equal(ctx.hysteresis(nm, lo, hi), w["edges"]), (tid, r, im.shape, "hysteresis")
                got = ctx.canny_batch(batch, sigma, lo, hi)
                assert np.array_equal(got, want_batch), (tid, r, "batch")
    except BaseException as e:  # noqa: BLE001 -- reported by the main thread
        errors.append((tid, repr(e)))
        try:
            barrier.abort()
        except Exception:
            pass


@pytest.mark.parametrize("n_threads", [4])
def test_contexts_on_concurrent_host_threads(hip, n_threads):
    errors = []
    barrier = threading.Barrier(n_threads)
    threads = [threading.Thread(target=_worker, args=(hip, t, 3, errors, barrier)) for t in range(n_threads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in threads), "a worker thread is still running"
    assert not errors, errors


def test_module_level_entry_points_from_threads(hip):
    """The reference-named functions of the binding (capi.canny, ...) run on a default context per calling thread,
    like the C++ shim's thread_local one."""
    img = synth_frame(200, 300, 77)
    want = oracle.canny(img, 1.4, 50, 150)
    errors = []

    def run():
        try:
            for _ in range(5):
                assert np.array_equal(hip.canny(img, 1.4, 50, 150), want)
        except BaseException as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=run) for _ in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
