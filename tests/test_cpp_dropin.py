"""The C++ drop-in layer (include/utils.h, include/cuda.h, libcanny_utils.so) and the CLI.

* building tests/cpp/test_utils_dropin.cpp against the drop-in headers proves the reference's own test
  source would compile against them (same names, same reference-to-pointer signatures) -- CPU, no GPU needed;
* running it on the GPU replays the reference's known-answer vectors through the C++ names with the
  reference's new[]/delete[] ownership;
* `Main` is run end to end on a PGM frame and its output compared with the oracle.
"""
import os
import subprocess

import numpy as np
import pytest

import oracle
from canny_edge_amd.synth import synth_frame

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "canny_edge_amd")
EXE = os.path.join(ROOT, "tests", "cpp", "test_utils_dropin")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _build():
    src = os.path.join(ROOT, "tests", "cpp", "test_utils_dropin.cpp")
    deps = [src, os.path.join(ROOT, "include", "utils.h"), os.path.join(ROOT, "include", "cuda.h"),
            os.path.join(ROOT, "include", "canny_frames.h"),
            os.path.join(PKG, "libcanny_utils.so")]
    if os.path.exists(EXE) and all(os.path.getmtime(EXE) >= os.path.getmtime(d) for d in deps):
        return
    subprocess.check_call([HIPCC, "-std=c++14", "-O1", "-x", "c++", src, "-x", "none", "-I" + os.path.join(ROOT, "include"),
                           "-L" + PKG, "-lcanny_utils", "-lcanny_hip", "-Wl,-rpath," + PKG, "-o", EXE])


def test_reference_test_source_shape_compiles_against_dropin_headers():
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_reference_vectors_through_cpp_dropin():
    _build()
    r = subprocess.run([EXE, os.path.join(ROOT, "tests", "golden", "test_gray_256x256.u8"),
                        os.path.join(ROOT, "tests", "golden", "test.jpg")], capture_output=True,
                       text=True, cwd=ROOT, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failures" in r.stdout


def _write_pgm(path, img):
    with open(path, "wb") as f:
        f.write(b"P5\n# written by the test-suite\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(img.tobytes())


def _read_pgm(path):
    data = open(path, "rb").read()
    parts = data.split(b"\n", 3)
    assert parts[0] == b"P5"
    w, h = map(int, parts[1].split())
    assert parts[2] == b"255"
    return np.frombuffer(parts[3], np.uint8, count=w * h).reshape(h, w)


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [[], ["-c"], ["-s"], ["-c", "-s"]])
def test_cli_end_to_end(tmp_path, flags):
    """./Main sigma minVal maxVal [-c] [-s] on a PGM frame: the written edge map equals the oracle's."""
    img = synth_frame(480, 640, 5)
    _write_pgm(tmp_path / "frame.pgm", img)
    exe = os.path.join(PKG, "Main")
    # positionals interleaved with flags, as the reference's parser allows (src/main.cpp:29-46)
    cmd = [exe, "1.4"] + flags + ["50", "-i", str(tmp_path / "frame.pgm"), "150", "-o", str(tmp_path)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Execution time:" in r.stdout
    got = _read_pgm(tmp_path / "canny_edges.pgm")
    want = oracle.canny(img, 1.4, 50, 150, stages=True)
    assert np.array_equal(got, want["edges"].astype(np.uint8))       # {0,255} survives min-max normalisation
    if "-s" in flags:
        # the reference shows each intermediate through normalize(..., 0, 255, NORM_MINMAX) + convertTo(CV_8U)
        # (src/utils.cpp:440-475): gaussian, gradient magnitude, suppressed magnitude.  OpenCV is not in this image,
        # so the min-max stretch itself is the documented formula, not OpenCV's code ("parity unpinned" for the
        # display scaling only; the planes underneath are the oracle's, bit for bit).
        def norm(plane):
            ref = plane.astype(np.float64)
            return np.rint((ref - ref.min()) * (255.0 / (ref.max() - ref.min()))).astype(np.uint8)

        for fname, key in (("canny_step1_gaussian.pgm", "smoothed"), ("canny_step2_gradient.pgm", "magnitude"),
                           ("canny_step3_nonmaximal.pgm", "nms")):
            got_step = _read_pgm(tmp_path / fname)
            assert np.array_equal(got_step, norm(want[key])), fname


@pytest.mark.gpu
def test_cli_batch_directory(tmp_path):
    """./Main sigma minVal maxVal -b dir: every PGM of the directory in one batch, 8-bit edge maps written back."""
    src, dst = tmp_path / "in", tmp_path / "out"
    src.mkdir()
    dst.mkdir()
    frames = {f"f{i:02d}": synth_frame(240, 328, 20 + i) for i in range(5)}
    for name, img in frames.items():
        _write_pgm(src / f"{name}.pgm", img)
    r = subprocess.run([os.path.join(PKG, "Main"), "1.0", "40", "120", "-b", str(src), "-o", str(dst)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Execution time:" in r.stdout and "5 frames of 328x240" in r.stdout
    for name, img in frames.items():
        got = _read_pgm(dst / f"{name}_edges.pgm")
        assert np.array_equal(got, oracle.canny(img, 1.0, 40, 120).astype(np.uint8)), name


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [[], ["-c"]])
def test_cli_reads_the_reference_jpeg(tmp_path, flags, luma_image, oracle_hashes):
    """./Main on the reference's own tests/test.jpg: decoded as cv::imread(IMREAD_GRAYSCALE) decodes it
    (include/canny_frames.h), the edge map equals the oracle's on the committed luminance plane."""
    cmd = [os.path.join(PKG, "Main"), "1.0", "50", "150", "-i", os.path.join(ROOT, "tests", "golden", "test.jpg"),
           "-o", str(tmp_path)] + flags
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    got = _read_pgm(tmp_path / "canny_edges.pgm")
    assert np.array_equal(got, oracle.canny(luma_image, 1.0, 50, 150).astype(np.uint8))
    assert int(np.count_nonzero(got)) == oracle_hashes["jpegluma256_s1.0_50_150"]["edge_pixels"]


@pytest.mark.gpu
def test_cli_batch_directory_mixes_jpeg_and_pgm(tmp_path, luma_image):
    import shutil
    src, dst = tmp_path / "in", tmp_path / "out"
    src.mkdir()
    dst.mkdir()
    other = synth_frame(256, 256, 77)
    shutil.copyfile(os.path.join(ROOT, "tests", "golden", "test.jpg"), src / "a_photo.JPG")
    _write_pgm(src / "b_card.pgm", other)
    (src / "notes.txt").write_text("not a frame")
    r = subprocess.run([os.path.join(PKG, "Main"), "1.0", "50", "150", "-b", str(src), "-o", str(dst)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "2 frames of 256x256" in r.stdout
    assert np.array_equal(_read_pgm(dst / "a_photo_edges.pgm"), oracle.canny(luma_image, 1.0, 50, 150).astype(np.uint8))
    assert np.array_equal(_read_pgm(dst / "b_card_edges.pgm"), oracle.canny(other, 1.0, 50, 150).astype(np.uint8))


def test_cli_reports_a_damaged_jpeg(tmp_path):
    """No GPU needed: the frame is rejected before any context exists."""
    data = open(os.path.join(ROOT, "tests", "golden", "test.jpg"), "rb").read()
    (tmp_path / "cut.jpg").write_bytes(data[:5000])
    r = subprocess.run([os.path.join(PKG, "Main"), "1.0", "50", "150", "-i", str(tmp_path / "cut.jpg")],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode != 0
    assert "truncated or damaged" in r.stdout and "Failed to open" in r.stdout


@pytest.mark.gpu
def test_cli_png_output(tmp_path):
    """-p: every output of the run as PNG (8-bit gray), same pixels as the PGM run."""
    Image = pytest.importorskip("PIL.Image")
    img = synth_frame(200, 333, 9)
    _write_pgm(tmp_path / "frame.pgm", img)
    r = subprocess.run([os.path.join(PKG, "Main"), "1.4", "50", "150", "-s", "-p", "-i", str(tmp_path / "frame.pgm"),
                        "-o", str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    want = oracle.canny(img, 1.4, 50, 150)
    assert np.array_equal(np.asarray(Image.open(tmp_path / "canny_edges.png")), want.astype(np.uint8))
    for name in ("canny_step1_gaussian.png", "canny_step2_gradient.png", "canny_step3_nonmaximal.png"):
        assert Image.open(tmp_path / name).size == (333, 200)
    assert not list(tmp_path.glob("canny_*.pgm"))
