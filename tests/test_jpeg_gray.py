"""The JPEG frame source and the PGM/PNG sink (include/canny_frames.h, csrc/jpeg_gray.cpp, csrc/png_gray.cpp) -- CPU only.

The reference's tests read their frame with cv::imread("test.jpg", IMREAD_GRAYSCALE) (tests/utils/test_utils.cpp:49):
libjpeg's luminance plane.  The decoder must return exactly those bytes:

* tests/golden/test.jpg (the reference's own data file) -> tests/golden/test_luma_256x256.u8 (made with PIL's
  libjpeg, tests/golden/make_fixtures.py);
* files PIL writes here (sizes that are not MCU multiples, gray / 4:4:4 / 4:2:2 / 4:2:0, qualities 5..100, restart
  intervals, optimised Huffman tables) against PIL's own decode of them;
* streams put together by hand below -- non-interleaved scans, tables PIL never writes, fill bytes, 16-bit
  quantisation tables -- again against PIL's decode;
* flavours it does not handle are refused by name, and damaged input never crashes it.
"""
import ctypes
import io
import os
import re
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
LIB = os.path.join(ROOT, "canny_edge_amd", "libcanny_utils.so")

OK, ERR_ARG, ERR_FORMAT, ERR_UNSUPPORTED = 0, 1, 2, 3


@pytest.fixture(scope="module")
def lib():
    L = ctypes.CDLL(LIB)
    L.canny_frames_last_error.restype = ctypes.c_char_p
    for f in (L.canny_frames_jpeg_info, L.canny_frames_jpeg_decode_gray):
        f.restype = ctypes.c_int
    L.canny_frames_jpeg_info.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int),
                                         ctypes.POINTER(ctypes.c_int)]
    L.canny_frames_jpeg_decode_gray.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                                ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    return L


def decode(L, data: bytes):
    """(status, image or error text)"""
    h, w = ctypes.c_int(), ctypes.c_int()
    st = L.canny_frames_jpeg_info(data, len(data), ctypes.byref(h), ctypes.byref(w))
    if st:
        return st, L.canny_frames_last_error().decode()
    out = np.empty((h.value, w.value), np.uint8)
    st = L.canny_frames_jpeg_decode_gray(data, len(data), out.ctypes.data, out.size, ctypes.byref(h), ctypes.byref(w))
    if st:
        return st, L.canny_frames_last_error().decode()
    assert (h.value, w.value) == out.shape
    return OK, out


def pil_luma(data: bytes):
    Image = pytest.importorskip("PIL.Image")
    im = Image.open(io.BytesIO(data))
    im.draft("L", im.size)  # libjpeg out_color_space = JCS_GRAYSCALE, scale 1/1
    assert im.mode == "L"
    return np.asarray(im)


def test_header_symbols_exported():
    header = open(os.path.join(ROOT, "include", "canny_frames.h")).read()
    declared = sorted(set(re.findall(r"\b(canny_frames_[a-z0-9_]+)\s*\(", header)))
    assert declared == ["canny_frames_jpeg_decode_gray", "canny_frames_jpeg_info", "canny_frames_last_error",
                        "canny_frames_write_gray"]
    L = ctypes.CDLL(LIB)
    assert all(hasattr(L, name) for name in declared)


def test_reference_fixture_decodes_to_the_committed_luminance_plane(lib, luma_image):
    data = open(os.path.join(GOLDEN, "test.jpg"), "rb").read()
    st, img = decode(lib, data)
    assert st == OK, img
    assert img.shape == (256, 256)
    assert np.array_equal(img, luma_image)


def test_reference_fixture_against_pil_here(lib):
    data = open(os.path.join(GOLDEN, "test.jpg"), "rb").read()
    st, img = decode(lib, data)
    assert st == OK and np.array_equal(img, pil_luma(data))


def _picture(rng, h, w, channels):
    """Smooth gradients + blocks + noise: exercises long zero runs as well as dense blocks."""
    yy, xx = np.mgrid[0:h, 0:w]
    planes = []
    for c in range(channels):
        p = 96 + 60 * np.sin(xx / (7.0 + c)) + 50 * np.cos(yy / (11.0 - c)) + rng.integers(-30, 31, (h, w))
        p[h // 3: h // 3 + max(1, h // 4), w // 4: w // 4 + max(1, w // 3)] = 255 * (c % 2)
        planes.append(np.clip(p, 0, 255).astype(np.uint8))
    return planes[0] if channels == 1 else np.dstack(planes)


@pytest.mark.parametrize("h,w", [(1, 1), (8, 8), (7, 13), (16, 16), (17, 33), (100, 75), (255, 257), (480, 640)])
def test_files_written_by_pil(lib, h, w):
    Image = pytest.importorskip("PIL.Image")
    from PIL import ImageFile
    ImageFile.MAXBLOCK = max(ImageFile.MAXBLOCK, 1 << 24)  # optimize=True needs the whole file in one encoder buffer
    rng = np.random.default_rng(h * 1000 + w)
    n = 0
    for channels in (1, 3):
        for quality in (5, 50, 90, 100):
            for subsampling in ((0, 1, 2) if channels == 3 else (0,)):
                for restart in (0, 1, 5):
                    for optimize in (False, True):
                        pic = _picture(rng, h, w, channels) if quality != 100 else \
                            rng.integers(0, 256, (h, w, 3) if channels == 3 else (h, w), dtype=np.uint8)
                        kw = {"quality": quality, "optimize": optimize}
                        if channels == 3:
                            kw["subsampling"] = subsampling
                        if restart:
                            kw["restart_marker_blocks"] = restart
                        bio = io.BytesIO()
                        Image.fromarray(pic).save(bio, "JPEG", **kw)
                        data = bio.getvalue()
                        st, img = decode(lib, data)
                        assert st == OK, (img, channels, quality, subsampling, restart, optimize)
                        assert np.array_equal(img, pil_luma(data)), (channels, quality, subsampling, restart, optimize)
                        n += 1
    assert n == 96


# ---- streams assembled by hand ----------------------------------------------------------------------------------
ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14,
          21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60,
          61, 54, 47, 55, 62, 63]
DC_SYMBOLS = list(range(12))                                                   # 12 codes of 4 bits
AC_SYMBOLS = [0x00, 0xF0] + [(r << 4) | s for r in range(16) for s in range(1, 11)]  # 162 codes of 8 bits


class Bits:
    def __init__(self):
        self.out = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, value, length):
        self.acc = (self.acc << length) | (value & ((1 << length) - 1))
        self.n += length
        while self.n >= 8:
            byte = (self.acc >> (self.n - 8)) & 255
            self.out.append(byte)
            if byte == 0xFF:
                self.out.append(0)
            self.n -= 8

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)  # pad with ones


def _magnitude(v):
    size = int(abs(v)).bit_length()
    return size, (v if v >= 0 else v + (1 << size) - 1)


def _encode_block(bits, zz, pred):
    """zz: 64 quantised coefficients in zig-zag order.  DC codes are 4 bits (index in DC_SYMBOLS), AC codes 8 bits."""
    size, extra = _magnitude(zz[0] - pred)
    bits.put(DC_SYMBOLS.index(size), 4)
    if size:
        bits.put(extra, size)
    run = 0
    last = max([k for k in range(1, 64) if zz[k]], default=0)
    for k in range(1, last + 1):
        if zz[k] == 0:
            run += 1
            continue
        while run > 15:
            bits.put(AC_SYMBOLS.index(0xF0), 8)
            run -= 16
        size, extra = _magnitude(zz[k])
        bits.put(AC_SYMBOLS.index((run << 4) | size), 8)
        bits.put(extra, size)
        run = 0
    if last < 63:
        bits.put(AC_SYMBOLS.index(0x00), 8)
    return zz[0]


def _segment(marker, payload):
    return bytes([0xFF, marker]) + struct.pack(">H", len(payload) + 2) + payload


def handmade_jpeg(rng, h, w, sampling, interleaved, restart=0, amplitude=40, density=0.15, wide_tables=False,
                  fill_bytes=False):
    """A sequential Huffman JPEG of random quantised coefficients.  sampling: [(h, v)] per component (1 or 3 of them)."""
    ncomp = len(sampling)
    hmax = max(s[0] for s in sampling)
    vmax = max(s[1] for s in sampling)
    if ncomp == 1:
        hmax = vmax = 1
    qt = rng.integers(1, 24, 64) if not wide_tables else rng.integers(200, 400, 64)
    out = bytearray(b"\xFF\xD8")
    out += _segment(0xE0, b"JFIF\0\x01\x01\0\0\x01\0\x01\0\0")
    if wide_tables:
        out += _segment(0xDB, bytes([0x10]) + b"".join(struct.pack(">H", int(q)) for q in qt))  # 16-bit entries
    else:
        out += _segment(0xDB, bytes([0x00]) + bytes(int(q) for q in qt))
    sof = struct.pack(">BHHB", 8, h, w, ncomp)
    for i, (sh, sv) in enumerate(sampling):
        sof += bytes([i + 1, (sh << 4) | sv, 0])
    out += _segment(0xC1 if wide_tables else 0xC0, sof)
    dc_counts = [0, 0, 0, 12] + [0] * 12
    ac_counts = [0] * 7 + [162] + [0] * 8
    out += _segment(0xC4, bytes([0x00]) + bytes(dc_counts) + bytes(DC_SYMBOLS))
    out += _segment(0xC4, bytes([0x10]) + bytes(ac_counts) + bytes(AC_SYMBOLS))
    if restart:
        out += _segment(0xDD, struct.pack(">H", restart))

    def block():
        zz = [0] * 64
        zz[0] = int(rng.integers(-amplitude, amplitude + 1))
        for k in range(1, 64):
            if rng.random() < density / (1 + k / 8):
                zz[k] = int(rng.integers(-amplitude, amplitude + 1))
        return zz

    def scan(comp_ids, blocks_of_mcu, n_mcu):
        hdr = bytes([len(comp_ids)]) + b"".join(bytes([c + 1, 0x00]) for c in comp_ids) + bytes([0, 63, 0])
        seg = bytearray(_segment(0xDA, hdr))
        bits = Bits()
        pred = {c: 0 for c in comp_ids}
        rst = 0
        for m in range(n_mcu):
            if restart and m and m % restart == 0:
                bits.flush()
                seg += bits.out
                if fill_bytes:
                    seg += b"\xFF\xFF"
                seg += bytes([0xFF, 0xD0 + rst])
                rst = (rst + 1) & 7
                bits = Bits()
                pred = {c: 0 for c in comp_ids}
            for c in comp_ids:
                for _ in range(blocks_of_mcu[c]):
                    pred[c] = _encode_block(bits, block(), pred[c])
        bits.flush()
        seg += bits.out
        return bytes(seg)

    mcux = -(-w // (8 * hmax))
    mcuy = -(-h // (8 * vmax))
    if interleaved and ncomp > 1:
        out += scan(list(range(ncomp)), {c: sampling[c][0] * sampling[c][1] for c in range(ncomp)}, mcux * mcuy)
    else:
        for c in reversed(range(ncomp)):  # chroma scans first: the decoder has to walk over them
            sh, sv = (1, 1) if ncomp == 1 else sampling[c]
            cw, ch = -(-w * sh // hmax), -(-h * sv // vmax)
            out += scan([c], {c: 1}, (-(-cw // 8)) * (-(-ch // 8)))
            if fill_bytes:
                out += b"\xFF\xFF\xFF"
    out += b"\xFF\xD9"
    return bytes(out)


@pytest.mark.parametrize("sampling,interleaved", [
    ([(1, 1)], True), ([(1, 1), (1, 1), (1, 1)], True), ([(2, 2), (1, 1), (1, 1)], True), ([(2, 1), (1, 1), (1, 1)], True),
    ([(1, 2), (1, 1), (1, 1)], True), ([(4, 1), (1, 1), (2, 1)], True), ([(2, 2), (1, 1), (1, 1)], False),
    ([(1, 1), (1, 1), (1, 1)], False), ([(2, 1), (1, 1), (1, 1)], False)])
@pytest.mark.parametrize("restart", [0, 1, 3])
def test_handmade_streams(lib, sampling, interleaved, restart):
    rng = np.random.default_rng(len(sampling) * 100 + restart + 7 * interleaved + sampling[0][0] * 13 + sampling[0][1])
    for (h, w) in [(8, 8), (19, 45), (64, 40), (33, 130)]:
        data = handmade_jpeg(rng, h, w, sampling, interleaved, restart, fill_bytes=(restart == 3))
        st, img = decode(lib, data)
        assert st == OK, (img, h, w)
        assert np.array_equal(img, pil_luma(data)), (h, w)


def test_sixteen_bit_quantisation_tables_and_saturation(lib):
    rng = np.random.default_rng(5)
    # entries of 200..400 on coefficients of +-4 drive the inverse DCT well outside [0, 255] (beyond +-512 around
    # mid-gray, where libjpeg's table lookup and its SIMD code differ): results saturate.  Larger coefficients
    # overflow the 16-bit workspace of libjpeg's SIMD code -- no flavour of libjpeg agrees with another there.
    data = handmade_jpeg(rng, 40, 56, [(2, 2), (1, 1), (1, 1)], True, wide_tables=True, density=0.5, amplitude=4)
    st, img = decode(lib, data)
    assert st == OK, img
    want = pil_luma(data)
    assert np.array_equal(img, want)
    assert (want == 0).any() and (want == 255).any()


def test_unsupported_flavours_are_named(lib):
    Image = pytest.importorskip("PIL.Image")
    pic = _picture(np.random.default_rng(3), 40, 40, 3)
    bio = io.BytesIO()
    Image.fromarray(pic).save(bio, "JPEG", progressive=True)
    st, msg = decode(lib, bio.getvalue())
    assert st == ERR_UNSUPPORTED and "progressive" in msg
    bio = io.BytesIO()
    Image.fromarray(np.dstack([pic, pic[:, :, 0]]), "CMYK").save(bio, "JPEG")
    st, msg = decode(lib, bio.getvalue())
    assert st == ERR_UNSUPPORTED and "4 components" in msg
    bio = io.BytesIO()
    Image.fromarray(pic).save(bio, "JPEG", keep_rgb=True)   # Adobe marker with transform 0: R, G, B planes
    st, msg = decode(lib, bio.getvalue())
    assert st == ERR_UNSUPPORTED and "RGB" in msg


def test_bad_arguments_and_foreign_files(lib):
    h, w = ctypes.c_int(), ctypes.c_int()
    assert lib.canny_frames_jpeg_info(None, 0, ctypes.byref(h), ctypes.byref(w)) == ERR_ARG
    assert decode(lib, b"P5\n2 2\n255\n\0\0\0\0")[0] == ERR_FORMAT
    assert decode(lib, b"\xFF\xD8")[0] == ERR_FORMAT
    assert decode(lib, b"\xFF\xD8\xFF\xD9")[0] == ERR_FORMAT
    data = open(os.path.join(GOLDEN, "test.jpg"), "rb").read()
    small = np.empty(100, np.uint8)
    st = lib.canny_frames_jpeg_decode_gray(data, len(data), small.ctypes.data, small.size, ctypes.byref(h), ctypes.byref(w))
    assert st == ERR_ARG and b"256x256" in lib.canny_frames_last_error()


def test_damaged_streams_do_not_crash(lib):
    """Truncations and byte flips: a fault or a hang is not acceptable; a file cut inside its scan is reported as damaged
    (libjpeg would pad it with gray and warn -- there is no warning channel here, so it is an error)."""
    data = open(os.path.join(GOLDEN, "test.jpg"), "rb").read()
    st, full = decode(lib, data)
    assert st == OK
    for cut in (3, 20, 200, 700, 1024, len(data) // 2, len(data) - 3):
        st, img = decode(lib, data[:cut])
        assert st == ERR_FORMAT, (cut, st)
    for cut in (len(data) - 2, len(data) - 1):          # only the end-of-image marker is missing: all the data is there
        st, img = decode(lib, data[:cut])
        assert st == OK and np.array_equal(img, full)
    st, msg = decode(lib, data[: len(data) // 2])
    assert st == ERR_FORMAT and "truncated" in msg
    gray = handmade_jpeg(np.random.default_rng(2), 64, 64, [(1, 1)], True)       # no restart markers in this one
    assert decode(lib, gray)[0] == OK
    st, msg = decode(lib, gray[: len(gray) - 40])
    assert st == ERR_FORMAT and "ends early" in msg
    rng = np.random.default_rng(11)
    for _ in range(400):
        b = bytearray(data)
        for _ in range(int(rng.integers(1, 6))):
            b[int(rng.integers(2, len(b)))] = int(rng.integers(0, 256))
        st, _img = decode(lib, bytes(b))
        assert st in (OK, ERR_ARG, ERR_FORMAT, ERR_UNSUPPORTED)


def test_python_frame_source(tmp_path, luma_image):
    from canny_edge_amd import frames
    assert np.array_equal(frames.imread_gray(os.path.join(GOLDEN, "test.jpg")), luma_image)
    (tmp_path / "f.pgm").write_bytes(b"P5\n# comment\n3 2\n255\n" + bytes(range(6)))
    assert np.array_equal(frames.imread_gray(str(tmp_path / "f.pgm")), np.arange(6, dtype=np.uint8).reshape(2, 3))
    with pytest.raises(frames.FrameError) as ei:
        frames.jpeg_decode_gray(b"\xff\xd8\xff\xd9")
    assert ei.value.status == ERR_FORMAT


# ---- the frame sink ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("h,w", [(1, 1), (3, 5), (256, 256), (300, 70000 // 300 + 7), (1100, 1000)])
def test_png_and_pgm_sink(tmp_path, h, w):
    """canny_frames_write_gray: the PNG is read back with PIL and, independently, taken apart by hand (chunk CRCs, zlib
    stream of stored blocks); sizes straddle the 65535-byte block and the 1 MiB IDAT limits."""
    import zlib
    L = ctypes.CDLL(LIB)
    L.canny_frames_write_gray.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    img = np.random.default_rng(h * w).integers(0, 256, (h, w), dtype=np.uint8)
    png, pgm = tmp_path / "frame.PNG", tmp_path / "frame.pgm"
    assert L.canny_frames_write_gray(str(png).encode(), img.ctypes.data, h, w) == OK
    assert L.canny_frames_write_gray(str(pgm).encode(), img.ctypes.data, h, w) == OK
    assert pgm.read_bytes() == b"P5\n%d %d\n255\n" % (w, h) + img.tobytes()
    data = png.read_bytes()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, kinds = 8, b"", []
    while pos < len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(kind + body)
        kinds.append(kind)
        if kind == b"IHDR":
            assert body == struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)
        if kind == b"IDAT":
            assert n <= 1 << 20
            idat += body
        pos += 12 + n
    assert kinds[0] == b"IHDR" and kinds[-1] == b"IEND" and set(kinds[1:-1]) == {b"IDAT"}
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, w + 1)
    assert not raw[:, 0].any() and np.array_equal(raw[:, 1:], img)
    Image = pytest.importorskip("PIL.Image")
    back = Image.open(str(png))
    assert back.mode == "L" and np.array_equal(np.asarray(back), img)
    assert L.canny_frames_write_gray(str(tmp_path / "no_such_dir" / "x.png").encode(), img.ctypes.data, h, w) == ERR_ARG
