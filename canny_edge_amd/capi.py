"""ctypes binding of ``libcanny_hip.so`` (C ABI in ``include/canny_hip.h``).

Two layers:

* :class:`Context` -- one GPU + one stream; methods take numpy arrays (host API) or raw device
  pointers as ints (``dev_*`` API, asynchronous on the context's stream).
* module-level functions named exactly like the reference's stage functions
  (``src/utils.h:8-22``: ``gaussian``, ``createGaussianKernel``, ``calculateXYGradient``,
  ``sobelOperator``, ``nonmaximalSuppression``, ``hysteresis``, ``findEdgePixels``, ``canny``) so that
  parity tests read like the reference's own tests.  They return new arrays instead of writing through
  reference-to-pointer out-parameters.

There is no CPU fallback anywhere in this module: if the shared library is missing, or no HIP device
is present, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CANNY_HIP_LIB: another build of the same library (A/B measurements of kernel variants, tools/ab_stage_times.py)
LIB_PATH = os.environ.get("CANNY_HIP_LIB") or os.path.join(_HERE, "libcanny_hip.so")

OK = 0
STAGE_GAUSSIAN, STAGE_SOBEL_NMS, STAGE_HYST_CLASSIFY, STAGE_HYST_PROPAGATE, STAGE_HYST_FINALIZE, \
    STAGE_SOBEL, STAGE_NMS, STAGE_XY_GRADIENT = range(8)
STAGE_NAMES = ("gaussian", "sobel_nms", "hyst_classify", "hyst_propagate", "hyst_finalize", "sobel", "nms",
               "xy_gradient")

# every symbol include/canny_hip.h declares (checked by tests/test_abi.py)
EXPORTS = (
    "canny_hip_version", "canny_hip_status_string", "canny_hip_device_count", "canny_hip_ctx_create",
    "canny_hip_ctx_destroy", "canny_hip_ctx_set_stream", "canny_hip_ctx_device", "canny_hip_ctx_set_option", "canny_hip_synchronize",
    "canny_hip_last_error", "canny_hip_last_hysteresis_iterations", "canny_hip_malloc", "canny_hip_free",
    "canny_hip_host_alloc", "canny_hip_host_free", "canny_hip_memcpy_h2d", "canny_hip_memcpy_d2h",
    "canny_hip_gaussian_kernel", "canny_hip_gaussian", "canny_hip_xy_gradient", "canny_hip_sobel", "canny_hip_nms",
    "canny_hip_hysteresis", "canny_hip_find_edge_pixels", "canny_hip_canny", "canny_hip_canny_batch",
    "canny_hip_canny_batch_u8", "canny_hip_dev_canny_u8", "canny_hip_canny_multi_gpu", "canny_hip_shard_range", "canny_hip_dev_gaussian", "canny_hip_dev_xy_gradient",
    "canny_hip_dev_sobel", "canny_hip_dev_nms", "canny_hip_dev_sobel_nms", "canny_hip_dev_hysteresis",
    "canny_hip_dev_canny", "canny_hip_dev_canny_stream", "canny_hip_dev_canny_stream_flush", "canny_hip_profile_enable", "canny_hip_profile_reset", "canny_hip_profile_get",
    "canny_hip_selftest_mag_angle", "canny_hip_selftest_div", "canny_hip_selftest_div_fma",
    "canny_hip_selftest_div_fma_table", "canny_hip_canny_multi_gpu_u8", "canny_hip_multi_gpu_set_option",
    "canny_hip_multi_gpu_release", "canny_hip_device_local_cpus", "canny_hip_selftest_cpulist_count",
    "canny_hip_dev_gaussian_u8", "canny_hip_dev_sobel_nms_u8in", "canny_hip_host_register", "canny_hip_host_unregister",
    "canny_hip_canny_batch_bits", "canny_hip_canny_multi_gpu_bits", "canny_hip_dev_canny_bits",
    "canny_hip_probe_copy", "canny_hip_ctx_get_option", "canny_hip_selftest_expand_bits",
    "canny_hip_selftest_march_order",
)

_lib: Optional[C.CDLL] = None


class CannyHipError(RuntimeError):
    def __init__(self, status: int, what: str, detail: str = ""):
        self.status = status
        msg = f"{what}: {status_string(status)}"
        if detail:
            msg += f" ({detail})"
        super().__init__(msg)


def load() -> C.CDLL:
    """Load the HIP library; fail loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(or `make -C canny_edge_amd/csrc`); there is no CPU fallback")
    # The batch pipeline's upload / compute / download streams want hardware queues of their own; HIP reads the variable
    # when its runtime initialises.  This module is host-application code (the library never touches the environment).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    L = C.CDLL(LIB_PATH)
    i, f, p, sz = C.c_int, C.c_float, C.c_void_p, C.c_size_t
    pp, ip = C.POINTER(C.c_void_p), C.POINTER(C.c_int)
    sig = {
        "canny_hip_version": ([], i),
        "canny_hip_status_string": ([i], C.c_char_p),
        "canny_hip_device_count": ([ip], i),
        "canny_hip_ctx_create": ([pp, i], i),
        "canny_hip_ctx_destroy": ([p], None),
        "canny_hip_ctx_set_stream": ([p, p], i),
        "canny_hip_ctx_device": ([p], i),
        "canny_hip_ctx_set_option": ([p, C.c_char_p, i], i),
        "canny_hip_synchronize": ([p], i),
        "canny_hip_last_error": ([p], C.c_char_p),
        "canny_hip_last_hysteresis_iterations": ([p], i),
        "canny_hip_malloc": ([p, pp, sz], i),
        "canny_hip_free": ([p, p], i),
        "canny_hip_host_alloc": ([p, pp, sz], i),
        "canny_hip_host_free": ([p, p], i),
        "canny_hip_host_register": ([p, p, sz], i),
        "canny_hip_host_unregister": ([p, p], i),
        "canny_hip_memcpy_h2d": ([p, p, p, sz], i),
        "canny_hip_memcpy_d2h": ([p, p, p, sz], i),
        "canny_hip_gaussian_kernel": ([f, p, i, ip], i),
        "canny_hip_gaussian": ([p, p, f, i, i, p], i),
        "canny_hip_xy_gradient": ([p, p, i, i, p, p], i),
        "canny_hip_sobel": ([p, p, i, i, p, p], i),
        "canny_hip_nms": ([p, p, p, i, i, p], i),
        "canny_hip_hysteresis": ([p, p, i, i, i, i], i),
        "canny_hip_find_edge_pixels": ([p, p, p, i, i, i, i, i], i),
        "canny_hip_canny": ([p, p, f, i, i, i, i, p], i),
        "canny_hip_canny_batch": ([p, p, i, f, i, i, i, i, p], i),
        "canny_hip_canny_batch_u8": ([p, p, i, f, i, i, i, i, p], i),
        "canny_hip_dev_canny_u8": ([p, p, f, i, i, i, i, i, p], i),
        "canny_hip_dev_canny_bits": ([p, p, f, i, i, i, i, i, p], i),
        "canny_hip_canny_batch_bits": ([p, p, i, f, i, i, i, i, p], i),
        "canny_hip_canny_multi_gpu_bits": ([p, i, f, i, i, i, i, p, i], i),
        "canny_hip_canny_multi_gpu": ([p, i, f, i, i, i, i, p, i], i),
        "canny_hip_canny_multi_gpu_u8": ([p, i, f, i, i, i, i, p, i], i),
        "canny_hip_multi_gpu_set_option": ([C.c_char_p, i], i),
        "canny_hip_multi_gpu_release": ([], i),
        "canny_hip_device_local_cpus": ([i, C.c_char_p, i], i),
        "canny_hip_selftest_cpulist_count": ([C.c_char_p], i),
        "canny_hip_shard_range": ([i, i, i, ip, ip], i),
        "canny_hip_dev_gaussian": ([p, p, f, i, i, i, p], i),
        "canny_hip_dev_xy_gradient": ([p, p, i, i, i, p, p], i),
        "canny_hip_dev_sobel": ([p, p, i, i, i, p, p], i),
        "canny_hip_dev_nms": ([p, p, p, i, i, i, p], i),
        "canny_hip_dev_sobel_nms": ([p, p, i, i, i, p], i),
        "canny_hip_dev_hysteresis": ([p, p, i, i, i, i, i], i),
        "canny_hip_dev_gaussian_u8": ([p, p, f, i, i, i, p], i),
        "canny_hip_dev_sobel_nms_u8in": ([p, p, i, i, i, p], i),
        "canny_hip_dev_canny": ([p, p, f, i, i, i, i, i, p], i),
        "canny_hip_dev_canny_stream": ([p, p, f, i, i, i, i, i, p], i),
        "canny_hip_dev_canny_stream_flush": ([p], i),
        "canny_hip_profile_enable": ([p, i], i),
        "canny_hip_profile_reset": ([p], i),
        "canny_hip_profile_get": ([p, i, C.POINTER(C.c_double), C.POINTER(C.c_long)], i),
        "canny_hip_probe_copy": ([p, p, p, C.c_size_t, i, C.POINTER(C.c_double)], i),
        "canny_hip_ctx_get_option": ([p, C.c_char_p, ip], i),
        "canny_hip_selftest_expand_bits": ([p, i, i, i, p, i], i),
        "canny_hip_selftest_march_order": ([i, i, p], i),
        "canny_hip_selftest_mag_angle": ([p, i, p, p], i),
        "canny_hip_selftest_div": ([p, f, C.POINTER(C.c_ulonglong), C.POINTER(C.c_float)], i),
        "canny_hip_selftest_div_fma": ([p, f, f, C.POINTER(C.c_ulonglong), C.POINTER(C.c_float)], i),
        "canny_hip_selftest_div_fma_table": ([i, C.POINTER(C.c_float), C.POINTER(C.c_float)], i),
    }
    for name, (args, res) in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = res
    _lib = L
    return L


def status_string(status: int) -> str:
    return load().canny_hip_status_string(status).decode()


def device_count() -> int:
    n = C.c_int(0)
    load().canny_hip_device_count(C.byref(n))
    return n.value


def expand_bits(bits: np.ndarray, height: int, width: int, u8: bool = False, threads: int = 4, out=None) -> np.ndarray:
    """Host-only: a packed bit map (rows MSB-first, padded to bytes) -> the short (or byte) edge plane, through the same
    thread pool the batch pipelines use for their compact transfer."""
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    assert bits.size == height * ((width + 7) // 8)
    if out is None:
        out = np.empty((height, width), np.uint8 if u8 else np.int16)
    st = load().canny_hip_selftest_expand_bits(bits.ctypes.data_as(C.c_void_p), height, width, int(u8),
                                               out.ctypes.data_as(C.c_void_p), threads)
    if st:
        raise CannyHipError(st, "selftest_expand_bits")
    return out


def march_order(n_segs: int, n_strips: int) -> np.ndarray:
    """Host-only: the (segment, strip) cell each wave index of a marching launch takes within a frame (border first)."""
    out = np.empty((n_segs * n_strips, 2), np.int32)
    st = load().canny_hip_selftest_march_order(n_segs, n_strips, out.ctypes.data_as(C.c_void_p))
    if st:
        raise CannyHipError(st, "selftest_march_order")
    return out


def fma_div_table():
    """The (divisor, c) pairs for which the Gaussian kernels replace a/divisor by fma(a, c, a)."""
    out, k = [], 0
    while True:
        s, c = C.c_float(0), C.c_float(0)
        if load().canny_hip_selftest_div_fma_table(k, C.byref(s), C.byref(c)):
            return out
        out.append((s.value, c.value))
        k += 1


def shard_range(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous frame range of shard ``rank`` of ``world`` (pure host logic, no GPU needed)."""
    b, e = C.c_int(0), C.c_int(0)
    st = load().canny_hip_shard_range(n_frames, rank, world, C.byref(b), C.byref(e))
    if st:
        raise CannyHipError(st, "canny_hip_shard_range")
    return b.value, e.value


def _hp(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _u8(img) -> np.ndarray:
    a = np.ascontiguousarray(img, dtype=np.uint8)
    if a.ndim != 2:
        raise ValueError("expected a 2-D uint8 image")
    return a


def _s16(img) -> np.ndarray:
    a = np.ascontiguousarray(img, dtype=np.int16)
    if a.ndim != 2:
        raise ValueError("expected a 2-D int16 plane")
    return a


class Context:
    """One GPU, one stream, reusable device workspaces (``canny_hip_ctx``)."""

    def __init__(self, device: int = 0):
        self._L = load()
        h = C.c_void_p()
        st = self._L.canny_hip_ctx_create(C.byref(h), device)
        if st:
            raise CannyHipError(st, "canny_hip_ctx_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            for p in getattr(self, "_pinned", []):
                self._L.canny_hip_host_free(self._h, C.c_void_p(p))
            self._pinned = []
            self._L.canny_hip_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, st: int, what: str):
        if st:
            raise CannyHipError(st, what, self._L.canny_hip_last_error(self._h).decode())

    # ---- plumbing ---------------------------------------------------------------------------
    @property
    def device(self) -> int:
        return self._L.canny_hip_ctx_device(self._h)

    def set_stream(self, hip_stream: int):
        """Launch on a caller-owned hipStream_t (e.g. ``torch.cuda.current_stream().cuda_stream``)."""
        self._check(self._L.canny_hip_ctx_set_stream(self._h, C.c_void_p(hip_stream)), "set_stream")

    def set_option(self, name: str, value: int):
        """Kernel-path selection ("gaussian_path" / "sobel_nms_path": 0 auto, 1 baseline, 2 wave-marching)."""
        self._check(self._L.canny_hip_ctx_set_option(self._h, name.encode(), value), f"set_option({name})")

    def get_option(self, name: str) -> int:
        v = C.c_int(0)
        self._check(self._L.canny_hip_ctx_get_option(self._h, name.encode(), C.byref(v)), f"get_option({name})")
        return v.value

    def synchronize(self):
        self._check(self._L.canny_hip_synchronize(self._h), "synchronize")

    def malloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._check(self._L.canny_hip_malloc(self._h, C.byref(p), nbytes), "malloc")
        return p.value

    def free(self, dptr: int):
        self._check(self._L.canny_hip_free(self._h, C.c_void_p(dptr)), "free")

    def h2d(self, dptr: int, host: np.ndarray):
        host = np.ascontiguousarray(host)
        self._check(self._L.canny_hip_memcpy_h2d(self._h, C.c_void_p(dptr), _hp(host), host.nbytes), "h2d")

    def d2h(self, host: np.ndarray, dptr: int):
        assert host.flags["C_CONTIGUOUS"]
        self._check(self._L.canny_hip_memcpy_d2h(self._h, _hp(host), C.c_void_p(dptr), host.nbytes), "d2h")

    @property
    def last_hysteresis_iterations(self) -> int:
        return self._L.canny_hip_last_hysteresis_iterations(self._h)

    def profile_enable(self, on: bool = True):
        self._check(self._L.canny_hip_profile_enable(self._h, int(on)), "profile_enable")

    def profile_reset(self):
        self._check(self._L.canny_hip_profile_reset(self._h), "profile_reset")

    def profile_get(self, stage: int) -> Tuple[float, int]:
        ms, n = C.c_double(0), C.c_long(0)
        self._check(self._L.canny_hip_profile_get(self._h, stage, C.byref(ms), C.byref(n)), "profile_get")
        return ms.value, n.value

    # ---- host-array stage API -----------------------------------------------------------------
    def gaussian(self, img, sigma: float) -> np.ndarray:
        a = _u8(img)
        out = np.empty(a.shape, np.int16)
        self._check(self._L.canny_hip_gaussian(self._h, _hp(a), sigma, a.shape[0], a.shape[1], _hp(out)), "gaussian")
        return out

    def xy_gradient(self, img) -> Tuple[np.ndarray, np.ndarray]:
        a = _s16(img)
        gx, gy = np.empty(a.shape, np.int16), np.empty(a.shape, np.int16)
        self._check(self._L.canny_hip_xy_gradient(self._h, _hp(a), a.shape[0], a.shape[1], _hp(gx), _hp(gy)),
                    "xy_gradient")
        return gx, gy

    def sobel(self, img) -> Tuple[np.ndarray, np.ndarray]:
        a = _s16(img)
        mag, ang = np.empty(a.shape, np.int16), np.empty(a.shape, np.int16)
        self._check(self._L.canny_hip_sobel(self._h, _hp(a), a.shape[0], a.shape[1], _hp(mag), _hp(ang)), "sobel")
        return mag, ang

    def nms(self, mag, ang) -> np.ndarray:
        m, a = _s16(mag), _s16(ang)
        if m.shape != a.shape:
            raise ValueError("magnitude/angle shape mismatch")
        out = np.empty(m.shape, np.int16)
        self._check(self._L.canny_hip_nms(self._h, _hp(m), _hp(a), m.shape[0], m.shape[1], _hp(out)), "nms")
        return out

    def hysteresis(self, cand, min_val: int, max_val: int) -> np.ndarray:
        c = _s16(cand).copy()
        self._check(self._L.canny_hip_hysteresis(self._h, _hp(c), c.shape[0], c.shape[1], min_val, max_val),
                    "hysteresis")
        return c

    def find_edge_pixels(self, cand, visited, start: int, min_val: int, max_val: int):
        c = _s16(cand).copy()
        v = np.ascontiguousarray(visited, dtype=np.uint8).copy()
        self._check(self._L.canny_hip_find_edge_pixels(self._h, _hp(c), _hp(v), start, min_val, max_val, c.shape[0],
                                                       c.shape[1]), "find_edge_pixels")
        return c, v

    def canny(self, img, sigma: float, min_val: int, max_val: int) -> np.ndarray:
        a = _u8(img)
        out = np.empty(a.shape, np.int16)
        self._check(self._L.canny_hip_canny(self._h, _hp(a), sigma, min_val, max_val, a.shape[0], a.shape[1],
                                            _hp(out)), "canny")
        return out

    def pinned_array(self, shape, dtype) -> np.ndarray:
        """numpy array over page-locked host memory (canny_hip_host_alloc); freed when the context closes.
        canny_batch DMA's pinned inputs/outputs in place instead of staging them."""
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        p = C.c_void_p()
        self._check(self._L.canny_hip_host_alloc(self._h, C.byref(p), nbytes), "host_alloc")
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(p.value)
        buf = (C.c_uint8 * nbytes).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def host_register(self, a: np.ndarray):
        """Page-lock an ordinary (C-contiguous) numpy array in place: canny_batch then DMA's it like pinned memory."""
        assert a.flags["C_CONTIGUOUS"]
        self._check(self._L.canny_hip_host_register(self._h, _hp(a), a.nbytes), "host_register")

    def host_unregister(self, a: np.ndarray):
        self._check(self._L.canny_hip_host_unregister(self._h, _hp(a)), "host_unregister")

    def canny_batch(self, imgs, sigma: float, min_val: int, max_val: int, out: Optional[np.ndarray] = None,
                    u8: bool = False, bits: bool = False) -> np.ndarray:
        """Host frames in, host edge maps out, transfers overlapped with the kernels.  u8=True returns the maps as
        uint8 (0 / 255) instead of the reference's int16: a third less PCIe traffic.  bits=True returns bit maps,
        uint8 [n_frames, H, (W + 7) // 8], rows packed MSB-first like numpy.packbits (see unpack_bits())."""
        a = np.ascontiguousarray(imgs, dtype=np.uint8)
        if a.ndim != 3:
            raise ValueError("expected uint8 [n_frames, H, W]")
        dtype = np.uint8 if (u8 or bits) else np.int16
        shape = bits_shape(a.shape) if bits else a.shape
        if out is None:
            out = np.empty(shape, dtype)
        elif out.shape != shape or out.dtype != dtype or not out.flags["C_CONTIGUOUS"]:
            raise ValueError(f"out must be a C-contiguous {np.dtype(dtype).name} array of shape {shape}")
        fn = self._L.canny_hip_canny_batch_bits if bits else \
            (self._L.canny_hip_canny_batch_u8 if u8 else self._L.canny_hip_canny_batch)
        self._check(fn(self._h, _hp(a), a.shape[0], sigma, min_val, max_val, a.shape[1], a.shape[2], _hp(out)),
                    "canny_batch")
        return out

    def selftest_mag_angle(self, lim: int = 1020):
        side = 2 * lim + 1
        mags = np.empty((side, side), np.int16)
        bins = np.empty((side, side), np.uint8)
        self._check(self._L.canny_hip_selftest_mag_angle(self._h, lim, _hp(mags), _hp(bins)), "selftest")
        return mags, bins

    def selftest_div(self, divisor: float) -> Tuple[int, float]:
        """(mismatches, largest mismatching dividend) of the Gaussian's reciprocal division against IEEE
        a/divisor over all floats a in [0,256]."""
        bad, worst = C.c_ulonglong(0), C.c_float(0.0)
        self._check(self._L.canny_hip_selftest_div(self._h, divisor, C.byref(bad), C.byref(worst)), "selftest_div")
        return bad.value, worst.value

    def selftest_div_fma(self, divisor: float, c: float) -> Tuple[int, float]:
        """Same for the one-instruction form fma(a, c, a)."""
        bad, worst = C.c_ulonglong(0), C.c_float(0.0)
        self._check(self._L.canny_hip_selftest_div_fma(self._h, divisor, c, C.byref(bad), C.byref(worst)),
                    "selftest_div_fma")
        return bad.value, worst.value

    # ---- device-pointer stage API (ints are device addresses; n_frames contiguous planes) -------
    def dev_gaussian(self, d_img: int, sigma: float, h: int, w: int, n: int, d_out: int):
        self._check(self._L.canny_hip_dev_gaussian(self._h, C.c_void_p(d_img), sigma, h, w, n, C.c_void_p(d_out)),
                    "dev_gaussian")

    def dev_xy_gradient(self, d_img: int, h: int, w: int, n: int, d_gx: int, d_gy: int):
        self._check(self._L.canny_hip_dev_xy_gradient(self._h, C.c_void_p(d_img), h, w, n, C.c_void_p(d_gx),
                                                      C.c_void_p(d_gy)), "dev_xy_gradient")

    def dev_sobel(self, d_img: int, h: int, w: int, n: int, d_mag: int, d_ang: int):
        self._check(self._L.canny_hip_dev_sobel(self._h, C.c_void_p(d_img), h, w, n, C.c_void_p(d_mag),
                                                C.c_void_p(d_ang)), "dev_sobel")

    def dev_nms(self, d_mag: int, d_ang: int, h: int, w: int, n: int, d_out: int):
        self._check(self._L.canny_hip_dev_nms(self._h, C.c_void_p(d_mag), C.c_void_p(d_ang), h, w, n,
                                              C.c_void_p(d_out)), "dev_nms")

    def dev_sobel_nms(self, d_smoothed: int, h: int, w: int, n: int, d_out: int):
        self._check(self._L.canny_hip_dev_sobel_nms(self._h, C.c_void_p(d_smoothed), h, w, n, C.c_void_p(d_out)),
                    "dev_sobel_nms")

    def probe_copy(self, d_src: int, d_dst: int, nbytes: int, launches: int = 10) -> float:
        """Average device milliseconds of a plain copy of nbytes (measurement aid, see canny_hip_probe_copy)."""
        ms = C.c_double(0)
        self._check(self._L.canny_hip_probe_copy(self._h, C.c_void_p(d_src), C.c_void_p(d_dst), nbytes, launches,
                                                 C.byref(ms)), "probe_copy")
        return ms.value

    def dev_gaussian_u8(self, d_img: int, sigma: float, h: int, w: int, n: int, d_out: int):
        """Gaussian storing the smoothed plane as bytes (the "smoothed_u8" path of canny())."""
        self._check(self._L.canny_hip_dev_gaussian_u8(self._h, C.c_void_p(d_img), sigma, h, w, n, C.c_void_p(d_out)),
                    "dev_gaussian_u8")

    def dev_sobel_nms_u8in(self, d_smoothed: int, h: int, w: int, n: int, d_out: int):
        self._check(self._L.canny_hip_dev_sobel_nms_u8in(self._h, C.c_void_p(d_smoothed), h, w, n, C.c_void_p(d_out)),
                    "dev_sobel_nms_u8in")

    def dev_hysteresis(self, d_cand: int, h: int, w: int, n: int, min_val: int, max_val: int):
        self._check(self._L.canny_hip_dev_hysteresis(self._h, C.c_void_p(d_cand), h, w, n, min_val, max_val),
                    "dev_hysteresis")

    def dev_canny(self, d_img: int, sigma: float, min_val: int, max_val: int, h: int, w: int, n: int, d_edges: int):
        self._check(self._L.canny_hip_dev_canny(self._h, C.c_void_p(d_img), sigma, min_val, max_val, h, w, n,
                                                C.c_void_p(d_edges)), "dev_canny")

    def dev_canny_stream(self, d_img: int, sigma: float, min_val: int, max_val: int, h: int, w: int, n: int,
                         d_edges: int):
        """canny() for a stream of batches: returns with this batch's hysteresis sweeps in flight; d_edges is
        complete once the next dev_canny_stream call or dev_canny_stream_flush() has returned."""
        self._check(self._L.canny_hip_dev_canny_stream(self._h, C.c_void_p(d_img), sigma, min_val, max_val, h, w, n,
                                                       C.c_void_p(d_edges)), "dev_canny_stream")

    def dev_canny_stream_flush(self):
        self._check(self._L.canny_hip_dev_canny_stream_flush(self._h), "dev_canny_stream_flush")

    def dev_canny_bits(self, d_img: int, sigma: float, min_val: int, max_val: int, h: int, w: int, n: int, d_bits: int):
        self._check(self._L.canny_hip_dev_canny_bits(self._h, C.c_void_p(d_img), sigma, min_val, max_val, h, w, n,
                                                     C.c_void_p(d_bits)), "dev_canny_bits")

    def dev_canny_u8(self, d_img: int, sigma: float, min_val: int, max_val: int, h: int, w: int, n: int, d_edges: int):
        self._check(self._L.canny_hip_dev_canny_u8(self._h, C.c_void_p(d_img), sigma, min_val, max_val, h, w, n,
                                                   C.c_void_p(d_edges)), "dev_canny_u8")


def bits_shape(frames_shape) -> Tuple[int, int, int]:
    """Shape of the bit maps of [n_frames, H, W] frames: every row padded to whole bytes."""
    n, h, w = frames_shape
    return (n, h, (w + 7) // 8)


def unpack_bits(bits: np.ndarray, width: int) -> np.ndarray:
    """Bit maps [n_frames, H, (W + 7) // 8] -> the reference's int16 edge maps (0 / 255)."""
    return np.unpackbits(bits, axis=-1)[..., :width].astype(np.int16) * 255


def canny_multi_gpu(imgs, sigma: float, min_val: int, max_val: int, n_devices: int = 0, u8: bool = False,
                    out: Optional[np.ndarray] = None, bits: bool = False) -> np.ndarray:
    """Shard [n_frames, H, W] by contiguous ranges over the node's GPUs (one host thread per GPU, each running
    the batch pipeline; per-device contexts are cached until multi_gpu_release())."""
    a = np.ascontiguousarray(imgs, dtype=np.uint8)
    if a.ndim != 3:
        raise ValueError("expected uint8 [n_frames, H, W]")
    dtype = np.uint8 if (u8 or bits) else np.int16
    shape = bits_shape(a.shape) if bits else a.shape
    if out is None:
        out = np.empty(shape, dtype)
    elif out.shape != shape or out.dtype != dtype or not out.flags["C_CONTIGUOUS"]:
        raise ValueError(f"out must be a C-contiguous {np.dtype(dtype).name} array of shape {shape}")
    fn = load().canny_hip_canny_multi_gpu_bits if bits else \
        (load().canny_hip_canny_multi_gpu_u8 if u8 else load().canny_hip_canny_multi_gpu)
    st = fn(_hp(a), a.shape[0], sigma, min_val, max_val, a.shape[1], a.shape[2], _hp(out), n_devices)
    if st:
        raise CannyHipError(st, "canny_multi_gpu")
    return out


def multi_gpu_set_option(name: str, value: int):
    st = load().canny_hip_multi_gpu_set_option(name.encode(), value)
    if st:
        raise CannyHipError(st, f"multi_gpu_set_option({name})")


def multi_gpu_release():
    load().canny_hip_multi_gpu_release()


def device_local_cpus(device: int) -> Optional[str]:
    """sysfs CPU list local to a GPU ("0-31,128-159"), None if the platform does not say."""
    buf = C.create_string_buffer(4096)
    st = load().canny_hip_device_local_cpus(device, buf, len(buf))
    return buf.value.decode() if st == 0 else None


def cpulist_count(text: str) -> int:
    return load().canny_hip_selftest_cpulist_count(text.encode())


# ---- the reference's stage names (src/utils.h:8-22) on a default context per calling thread --------
# (the reference's functions are re-entrant; a context serves one thread at a time, so -- like the C++ shim's
# thread_local context, csrc/utils_shim.cpp -- every thread that uses these names gets a context of its own)
_default = threading.local()


def default_context() -> Context:
    ctx = getattr(_default, "ctx", None)
    if ctx is None:
        ctx = _default.ctx = Context(int(os.environ.get("CANNY_HIP_DEVICE", "0")))
    return ctx


def createGaussianKernel(sigma: float) -> np.ndarray:
    """src/utils.cpp:77-95 (host-side): returns the normalised float taps; len() is the window."""
    taps = np.zeros(129, np.float32)
    w = C.c_int(0)
    st = load().canny_hip_gaussian_kernel(sigma, _hp(taps), taps.size, C.byref(w))
    if st:
        raise CannyHipError(st, "createGaussianKernel")
    return taps[:w.value].copy()


def gaussian(img, sigma: float) -> np.ndarray:
    return default_context().gaussian(img, sigma)


def calculateXYGradient(img):
    return default_context().xy_gradient(img)


def sobelOperator(img):
    return default_context().sobel(img)


def nonmaximalSuppression(grad, angle) -> np.ndarray:
    return default_context().nms(grad, angle)


def hysteresis(edgeCandidates, minVal: int, maxVal: int) -> np.ndarray:
    return default_context().hysteresis(edgeCandidates, minVal, maxVal)


def findEdgePixels(edgeCandidates, visited, start: int, minVal: int, maxVal: int):
    return default_context().find_edge_pixels(edgeCandidates, visited, start, minVal, maxVal)


def canny(img, sigma: float, minVal: int, maxVal: int) -> np.ndarray:
    return default_context().canny(img, sigma, minVal, maxVal)
