"""Frame sources for the hot path (include/canny_frames.h): what the reference gets from OpenCV.

``imread_gray(path)`` stands in for ``cv::imread(path, IMREAD_GRAYSCALE)`` (tests/utils/test_utils.cpp:49 of the
reference) for baseline JPEG files -- libjpeg's luminance plane, byte for byte -- and for binary PGM files.  Host code;
the decoder lives in ``libcanny_utils.so`` (csrc/jpeg_gray.cpp).
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcanny_utils.so")
_lib = None


class FrameError(RuntimeError):
    def __init__(self, status: int, text: str):
        super().__init__(f"{text} (canny_frames status {status})")
        self.status = status


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise FileNotFoundError(f"{_LIB_PATH} is not built: run __graft_entry__.build()")
        L = ctypes.CDLL(_LIB_PATH)
        L.canny_frames_last_error.restype = ctypes.c_char_p
        L.canny_frames_jpeg_info.restype = ctypes.c_int
        L.canny_frames_jpeg_info.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int),
                                             ctypes.POINTER(ctypes.c_int)]
        L.canny_frames_jpeg_decode_gray.restype = ctypes.c_int
        L.canny_frames_jpeg_decode_gray.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                                    ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        _lib = L
    return _lib


def jpeg_decode_gray(data: bytes) -> np.ndarray:
    """Baseline JPEG bytes -> [H, W] uint8, as IMREAD_GRAYSCALE returns them."""
    L = _load()
    h, w = ctypes.c_int(), ctypes.c_int()
    st = L.canny_frames_jpeg_info(data, len(data), ctypes.byref(h), ctypes.byref(w))
    if st:
        raise FrameError(st, L.canny_frames_last_error().decode())
    out = np.empty((h.value, w.value), np.uint8)
    st = L.canny_frames_jpeg_decode_gray(data, len(data), out.ctypes.data, out.size, ctypes.byref(h), ctypes.byref(w))
    if st:
        raise FrameError(st, L.canny_frames_last_error().decode())
    return out


def imread_gray(path: str) -> np.ndarray:
    """A JPEG (told by its first two bytes) or a binary PGM file as an [H, W] uint8 frame."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:2] == b"\xff\xd8":
        return jpeg_decode_gray(data)
    tokens, pos = [], 0
    while len(tokens) < 4:                      # "P5", width, height, maxval; '#' starts a comment
        while pos < len(data) and data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            pos = data.index(b"\n", pos) + 1
            continue
        end = pos
        while end < len(data) and not data[end:end + 1].isspace():
            end += 1
        tokens.append(data[pos:end])
        pos = end
    if tokens[0] != b"P5" or int(tokens[3]) > 255:
        raise FrameError(2, f"{path}: neither a JPEG nor an 8-bit binary PGM")
    w, h = int(tokens[1]), int(tokens[2])
    return np.frombuffer(data, np.uint8, count=w * h, offset=pos + 1).reshape(h, w).copy()
