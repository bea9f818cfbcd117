"""ctypes front-end for the CPU oracle (oracle/canny_oracle.c).

TEST INFRASTRUCTURE ONLY.  Import this from tests/, from ``__graft_entry__.smoke()`` and from
``bench.py``'s ``cpu_baseline`` leg -- never from ``canny_edge_amd`` (the product).  The functions
mirror the reference's stage functions (``src/utils.h:8-22`` of StevenChang5/Canny_Edge) with numpy
arrays in place of ``new[]`` buffers.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcanny_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the oracle with its contract flags (see oracle/Makefile)."""
    src = os.path.join(_HERE, "canny_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        i32, f32 = C.c_int, C.c_float
        p = C.c_void_p
        L.canny_oracle_gaussian_window.argtypes = [f32]
        L.canny_oracle_gaussian_window.restype = i32
        L.canny_oracle_gaussian_kernel.argtypes = [f32, p, i32]
        L.canny_oracle_gaussian_kernel.restype = i32
        L.canny_oracle_gaussian.argtypes = [p, f32, i32, i32, p, p]
        L.canny_oracle_gaussian.restype = i32
        L.canny_oracle_xy_gradient.argtypes = [p, i32, i32, p, p]
        L.canny_oracle_xy_gradient.restype = i32
        L.canny_oracle_angle_bin.argtypes = [i32, i32]
        L.canny_oracle_angle_bin.restype = i32
        L.canny_oracle_magnitude.argtypes = [i32, i32]
        L.canny_oracle_magnitude.restype = i32
        L.canny_oracle_sobel.argtypes = [p, i32, i32, p, p]
        L.canny_oracle_sobel.restype = i32
        L.canny_oracle_nms.argtypes = [p, p, i32, i32, p]
        L.canny_oracle_nms.restype = i32
        L.canny_oracle_find_edge_pixels.argtypes = [p, p, i32, i32, i32, i32, i32]
        L.canny_oracle_find_edge_pixels.restype = i32
        L.canny_oracle_hysteresis.argtypes = [p, i32, i32, i32, i32]
        L.canny_oracle_hysteresis.restype = i32
        L.canny_oracle_canny.argtypes = [p, f32, i32, i32, i32, i32, p, p, p, p, p, p]
        L.canny_oracle_canny.restype = i32
        L.canny_oracle_angle_table.argtypes = [i32, p]
        L.canny_oracle_angle_table.restype = None
        L.canny_oracle_magnitude_table.argtypes = [i32, p]
        L.canny_oracle_magnitude_table.restype = None
        _lib = L
    return _lib


def _ptr(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _check(rc: int, what: str):
    if rc != 0:
        raise ValueError(f"oracle {what} failed with status {rc}")


def _u8(img) -> np.ndarray:
    a = np.ascontiguousarray(img, dtype=np.uint8)
    if a.ndim != 2:
        raise ValueError("expected a 2-D image")
    return a


def _s16(img) -> np.ndarray:
    a = np.ascontiguousarray(img, dtype=np.int16)
    if a.ndim != 2:
        raise ValueError("expected a 2-D image")
    return a


def gaussian_kernel(sigma: float) -> np.ndarray:
    """createGaussianKernel (src/utils.cpp:77-95): normalised float taps, window 1+2*ceil(3*sigma)."""
    taps = np.zeros(1024, dtype=np.float32)
    w = lib().canny_oracle_gaussian_kernel(sigma, _ptr(taps), taps.size)
    if w < 0:
        raise ValueError(f"sigma {sigma} gives an unsupported window")
    return taps[:w].copy()


def gaussian(img, sigma: float, return_row_pass: bool = False):
    """gaussian (src/utils.cpp:26-68): u8 [H,W] -> int16 [H,W]."""
    a = _u8(img)
    h, w = a.shape
    out = np.empty((h, w), dtype=np.int16)
    tmp = np.empty((h, w), dtype=np.float32) if return_row_pass else None
    _check(lib().canny_oracle_gaussian(_ptr(a), sigma, h, w, _ptr(out), _ptr(tmp)), "gaussian")
    return (out, tmp) if return_row_pass else out


def xy_gradient(img):
    """calculateXYGradient (src/utils.cpp:106-187): int16 [H,W] -> (gx, gy)."""
    a = _s16(img)
    h, w = a.shape
    gx = np.empty((h, w), dtype=np.int16)
    gy = np.empty((h, w), dtype=np.int16)
    _check(lib().canny_oracle_xy_gradient(_ptr(a), h, w, _ptr(gx), _ptr(gy)), "xy_gradient")
    return gx, gy


def angle_bin(gx: int, gy: int) -> int:
    return lib().canny_oracle_angle_bin(int(gx), int(gy))


def magnitude(gx: int, gy: int) -> int:
    return lib().canny_oracle_magnitude(int(gx), int(gy))


def angle_table(lim: int = 1020) -> np.ndarray:
    """Angle bin for every (gx, gy) in [-lim, lim]^2; index [gy + lim, gx + lim]."""
    side = 2 * lim + 1
    t = np.empty((side, side), dtype=np.uint8)
    lib().canny_oracle_angle_table(lim, _ptr(t))
    return t


def magnitude_table(lim: int = 1020) -> np.ndarray:
    """Magnitude for every (gx, gy) in [-lim, lim]^2; index [gy + lim, gx + lim]."""
    side = 2 * lim + 1
    t = np.empty((side, side), dtype=np.int16)
    lib().canny_oracle_magnitude_table(lim, _ptr(t))
    return t


def sobel(img):
    """sobelOperator (src/utils.cpp:201-236): int16 [H,W] -> (magnitude, angle in {0,45,90,135})."""
    a = _s16(img)
    h, w = a.shape
    mag = np.empty((h, w), dtype=np.int16)
    ang = np.empty((h, w), dtype=np.int16)
    _check(lib().canny_oracle_sobel(_ptr(a), h, w, _ptr(mag), _ptr(ang)), "sobel")
    return mag, ang


def nms(mag, ang):
    """nonmaximalSuppression (src/utils.cpp:248-308)."""
    m, a = _s16(mag), _s16(ang)
    if m.shape != a.shape:
        raise ValueError("magnitude/angle shape mismatch")
    h, w = m.shape
    out = np.empty((h, w), dtype=np.int16)
    _check(lib().canny_oracle_nms(_ptr(m), _ptr(a), h, w, _ptr(out)), "nms")
    return out


def find_edge_pixels(cand, visited, start: int, min_val: int, max_val: int):
    """findEdgePixels (src/utils.cpp:360-427).  Returns new (cand, visited) copies."""
    c = _s16(cand).copy()
    v = np.ascontiguousarray(visited, dtype=np.uint8).copy()
    h, w = c.shape
    _check(lib().canny_oracle_find_edge_pixels(_ptr(c), _ptr(v), start, min_val, max_val, h, w), "find_edge_pixels")
    return c, v


def hysteresis(cand, min_val: int, max_val: int):
    """hysteresis (src/utils.cpp:322-342).  Returns a new array (the reference works in place)."""
    c = _s16(cand).copy()
    h, w = c.shape
    _check(lib().canny_oracle_hysteresis(_ptr(c), h, w, min_val, max_val), "hysteresis")
    return c


def canny(img, sigma: float, min_val: int, max_val: int, stages: bool = False):
    """canny (src/utils.cpp:429-492) without the display code.

    Returns the {0,255} int16 edge map, or with ``stages=True`` a dict holding every intermediate
    plane and the per-stage wall times (gaussian, sobel, nms, hysteresis, total) in seconds.
    """
    a = _u8(img)
    h, w = a.shape
    edges = np.empty((h, w), dtype=np.int16)
    if not stages:
        _check(lib().canny_oracle_canny(_ptr(a), sigma, min_val, max_val, h, w, _ptr(edges),
                                        None, None, None, None, None), "canny")
        return edges
    sm = np.empty((h, w), dtype=np.int16)
    mg = np.empty((h, w), dtype=np.int16)
    an = np.empty((h, w), dtype=np.int16)
    nm = np.empty((h, w), dtype=np.int16)
    secs = np.zeros(5, dtype=np.float64)
    _check(lib().canny_oracle_canny(_ptr(a), sigma, min_val, max_val, h, w, _ptr(edges), _ptr(sm),
                                    _ptr(mg), _ptr(an), _ptr(nm), _ptr(secs)), "canny")
    return {"edges": edges, "smoothed": sm, "magnitude": mg, "angle": an, "nms": nm,
            "seconds": dict(zip(("gaussian", "sobel", "nms", "hysteresis", "total"), secs.tolist()))}
