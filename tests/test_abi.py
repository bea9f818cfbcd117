"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/canny_hip.h declares, refuses to run without a GPU (no CPU fallback), and its host-only
pieces (Gaussian taps, shard ranges) match the oracle.  No kernel is launched here."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import oracle
from canny_edge_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu() -> bool:
    return capi.device_count() > 0


def test_header_symbols_all_exported():
    header = open(os.path.join(ROOT, "include", "canny_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(canny_hip_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 35
    lib = ctypes.CDLL(capi.LIB_PATH)
    missing = [name for name in declared if not hasattr(lib, name)]
    assert not missing, f"declared in canny_hip.h but not exported: {missing}"
    assert sorted(capi.EXPORTS) == declared, "capi.EXPORTS out of sync with the header"


def test_version_and_status_strings():
    L = capi.load()
    header = open(os.path.join(ROOT, "include", "canny_hip.h")).read()
    assert L.canny_hip_version() == int(re.search(r"#define CANNY_HIP_VERSION (\d+)", header).group(1)) >= 200
    assert capi.status_string(0) == "ok"
    assert "CPU fallback" in capi.status_string(3)


def test_no_cpu_fallback_without_gpu():
    if _has_gpu():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.CannyHipError) as ei:
        capi.Context(0)
    assert ei.value.status == 3  # CANNY_HIP_ERR_NO_DEVICE


@pytest.mark.parametrize("sigma", [0.3, 0.5, 0.8, 1.0, 1.4, 2.0, 2.5, 3.3, 7.0, 21.0])
def test_host_gaussian_taps_match_oracle_bitwise(sigma):
    got = capi.createGaussianKernel(sigma)
    want = oracle.gaussian_kernel(sigma)
    assert got.dtype == np.float32 and got.shape == want.shape
    assert got.tobytes() == want.tobytes()


def test_host_gaussian_taps_reference_vector(ref_vectors):
    case = ref_vectors["gaussian_kernel"][1]  # tests/utils/test_utils.cpp:21-32
    got = capi.createGaussianKernel(case["sigma"])
    assert len(got) == case["window"]
    eps = float(np.finfo(np.float32).eps)
    assert all(abs(np.float32(w) - g) < eps for g, w in zip(got, case["expected"]))
    k2 = capi.createGaussianKernel(2.0)       # tests/utils/test_utils.cpp:34-45
    assert len(k2) == 13 and all(k2[i] == k2[12 - i] for i in range(7))


@pytest.mark.parametrize("sigma", [0.0, -1.0, float("nan"), float("inf"), 30.0])
def test_host_gaussian_taps_rejects_bad_sigma(sigma):
    with pytest.raises(capi.CannyHipError):
        capi.createGaussianKernel(sigma)


@pytest.mark.parametrize("n,world", [(8192, 8), (1024, 3), (7, 8), (0, 4), (1, 1), (100, 6)])
def test_shard_ranges_partition_the_batch(n, world):
    ranges = [capi.shard_range(n, r, world) for r in range(world)]
    assert ranges[0][0] == 0 and ranges[-1][1] == n
    for (b0, e0), (b1, e1) in zip(ranges, ranges[1:]):
        assert e0 == b1 and b0 <= e0 and b1 <= e1
    sizes = [e - b for b, e in ranges]
    assert max(sizes) - min(sizes) <= 1


def test_utils_shim_exports_reference_mangled_names():
    """The C++ drop-in must export exactly the symbols the reference's utils.h / cuda.h produce
    (SURVEY.md 8b lists the mangled names of src/utils.h:8-22)."""
    so = os.path.join(ROOT, "canny_edge_amd", "libcanny_utils.so")
    assert os.path.exists(so)
    syms = subprocess.check_output(["nm", "-D", "--defined-only", so], text=True)
    for name in ("_Z8gaussianRPhfiiRPs", "_Z20createGaussianKernelRPffPi", "_Z19calculateXYGradientRPsiiS0_S0_",
                 "_Z13sobelOperatorRPsiiS0_S0_", "_Z21nonmaximalSuppressionRPsS0_iiS0_", "_Z10hysteresisRPsiiii",
                 "_Z14findEdgePixelsRPsRPbiiiii", "_Z5cannyPhfiiiib"):
        assert name in syms, name
    for name in ("cuda_gaussian", "cuda_sobel", "cuda_nonmaixmal_suppression", "cuda_canny"):
        assert re.search(rf"_Z\d+{name}", syms), name


def test_cli_keeps_reference_grammar():
    """src/main.cpp:29-76: three positionals anywhere among the flags, usage + exit(0) otherwise,
    max > min, both in [0,255], every failure exits with status 0."""
    exe = os.path.join(ROOT, "canny_edge_amd", "Main")
    r = subprocess.run([exe, "1.0", "50"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stderr.startswith("USAGE:")
    r = subprocess.run([exe, "-s", "1.0", "150", "-c", "50"], capture_output=True, text=True)
    assert r.returncode == 0 and "minVal must be less than maxVal" in r.stderr
    r = subprocess.run([exe, "1.0", "-5", "50"], capture_output=True, text=True)
    assert r.returncode == 0 and "minVal must be in the range of [0,255]" in r.stderr
    r = subprocess.run([exe, "1.0", "50", "300"], capture_output=True, text=True)
    assert r.returncode == 0 and "maxVal must be in the range of [0,255]" in r.stderr


@pytest.mark.parametrize("text,count", [("0-3,8,10-11", 7), ("0", 1), ("0-31,128-159\n", 64), ("5-5", 1), ("", 0),
                                        ("3-1", 0), ("a-b", 0), ("1,,2", 2)])
def test_cpulist_parser_behind_the_numa_binding(text, count):
    """canny_hip_canny_multi_gpu binds each shard's threads to the GPU's sysfs local_cpulist; the parser is host
    logic and testable without a GPU."""
    assert capi.cpulist_count(text) == count


def test_multi_gpu_options_are_validated_without_a_gpu():
    for name in ("tune_batch_workers", "tune_batch_chunk_mb", "tune_batch_chunk_frames", "tune_batch_pipe_mode",
                 "allow_device_reuse", "numa_affinity"):
        capi.multi_gpu_set_option(name, 1)
        capi.multi_gpu_set_option(name, 0)
    capi.multi_gpu_set_option("numa_affinity", 1)
    with pytest.raises(capi.CannyHipError):
        capi.multi_gpu_set_option("no_such_option", 1)
    with pytest.raises(capi.CannyHipError):
        capi.multi_gpu_set_option("allow_device_reuse", 2)
    capi.multi_gpu_release()  # nothing cached: a no-op


def test_public_headers_are_plain_c(tmp_path):
    """The boundary is a C ABI: canny_hip.h and canny_frames.h must compile as strict C99 (no C++ in the signatures),
    and a C program must link against the libraries with nothing but those headers."""
    src = tmp_path / "user.c"
    src.write_text('#include "canny_hip.h"\n#include "canny_frames.h"\n#include <stdio.h>\n'
                   'int main(void) {\n'
                   '    int n = -1;\n'
                   '    int st = canny_hip_device_count(&n);\n'
                   '    int h = 0, w = 0;\n'
                   '    int fr = canny_frames_jpeg_info("xx", 2, &h, &w);\n'
                   '    printf("%d %d %d %d\\n", canny_hip_version(), st, n >= 0, fr);\n'
                   '    return 0;\n}\n')
    exe = tmp_path / "user"
    pkg = os.path.join(ROOT, "canny_edge_amd")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           str(src), "-L" + pkg, "-lcanny_utils", "-lcanny_hip", "-Wl,-rpath," + pkg, "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    version, _status, counted, frames_status = out.stdout.split()
    assert int(version) >= 200 and counted == "1" and frames_status == "2"   # "xx" is not a JPEG: CANNY_FRAMES_ERR_FORMAT


@pytest.mark.parametrize("u8", [False, True], ids=["short_plane", "byte_plane"])
@pytest.mark.parametrize("shape", [(1, 1), (3, 7), (5, 8), (7, 9), (130, 64), (257, 203), (300, 3840), (2160, 131)])
def test_bit_map_expansion_of_the_compact_transfer(shape, u8):
    """Host logic of the batch pipelines (no GPU involved): the 1-bit edge map that crosses PCIe is expanded into the
    reference's short plane (EDGE = 255 / NOEDGE = 0, src/utils.h:5-6), or a byte plane, by the library's thread pool.
    Checked against numpy for widths that are and are not multiples of 8, one and several row blocks, unaligned output."""
    h, w = shape
    rng = np.random.default_rng(h * 4099 + w)
    plane = (rng.random((h, w)) < 0.3)
    bits = np.packbits(plane, axis=-1)                       # rows MSB-first, padded to bytes
    want = (plane * 255).astype(np.uint8 if u8 else np.int16)
    for threads in (1, 5):
        got = capi.expand_bits(bits, h, w, u8=u8, threads=threads)
        assert np.array_equal(got, want), (shape, u8, threads)
    # an output that starts at an odd element (no 16-byte alignment: the streaming-store path must not be taken)
    backing = np.full(h * w + 8, 77, want.dtype)
    out = backing[3:3 + h * w].reshape(h, w)
    capi.expand_bits(bits, h, w, u8=u8, threads=3, out=out)
    assert np.array_equal(out, want) and (backing[:3] == 77).all() and (backing[3 + h * w:] == 77).all()


def test_march_cell_order_is_a_bijection_with_the_border_first():
    """Host-only: the order in which marching waves take a frame's (segment, strip) cells (csrc/canny_kernels.h
    march_cell_of) covers every cell exactly once and puts the border cells -- the expensive instantiations -- first."""
    from canny_edge_amd import capi
    for n_segs, n_strips in [(1, 1), (1, 7), (7, 1), (2, 2), (2, 9), (9, 2), (3, 3), (15, 16), (36, 9), (5, 4), (100, 3)]:
        cells = capi.march_order(n_segs, n_strips)
        assert cells.shape == (n_segs * n_strips, 2)
        assert len({(int(g), int(s)) for g, s in cells}) == n_segs * n_strips
        assert cells[:, 0].min() == 0 and cells[:, 0].max() == n_segs - 1
        assert cells[:, 1].min() == 0 and cells[:, 1].max() == n_strips - 1
        border = (cells[:, 0] == 0) | (cells[:, 0] == n_segs - 1) | (cells[:, 1] == 0) | (cells[:, 1] == n_strips - 1)
        n_border = int(border.sum())
        assert border[:n_border].all() and not border[n_border:].any()
        if n_segs >= 3 and n_strips >= 3:  # interior cells stay row-major (neighbours share halo rows and columns)
            inner = cells[n_border:]
            assert np.array_equal(inner, np.array([(g, s) for g in range(1, n_segs - 1) for s in range(1, n_strips - 1)]))
