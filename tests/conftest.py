import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ref_vectors():
    """Known-answer vectors transcribed from the reference's tests/utils/test_utils.cpp."""
    with open(os.path.join(GOLDEN, "reference_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle_hashes():
    with open(os.path.join(GOLDEN, "oracle_stage_hashes.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def fixture_image():
    """The reference's tests/test.jpg decoded to 256x256 gray (see tests/golden/make_fixtures.py)."""
    return np.fromfile(os.path.join(GOLDEN, "test_gray_256x256.u8"), dtype=np.uint8).reshape(256, 256)


@pytest.fixture(scope="session")
def luma_image():
    """The same file's luminance plane as libjpeg returns it for JCS_GRAYSCALE -- what cv::imread(IMREAD_GRAYSCALE)
    gives the reference's tests (tests/utils/test_utils.cpp:49); made by tests/golden/make_fixtures.py with PIL."""
    return np.fromfile(os.path.join(GOLDEN, "test_luma_256x256.u8"), dtype=np.uint8).reshape(256, 256)


@pytest.fixture(scope="session")
def hip():
    """The product C-ABI binding; GPU tests fail loudly if it cannot be loaded."""
    from canny_edge_amd import capi
    capi.load()
    return capi
