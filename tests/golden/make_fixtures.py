#!/usr/bin/env python3
"""Regenerates the data fixtures under tests/golden/ (run in the build container only).

1. test_gray_256x256.u8 -- the reference's only image fixture (tests/test.jpg, 256x256 JPEG, used by
   tests/utils/test_utils.cpp:47-104) decoded to 8-bit gray with PIL (``convert('L')``) and stored as
   65,536 raw bytes, row-major.  The reference decodes with cv::imread(IMREAD_GRAYSCALE); OpenCV is
   not installed here, so the decode may differ from OpenCV's by +-1 LSB.  The three reference tests
   that use the image only assert "sum != 0" and "all values in [0,255]", which do not depend on that.
   test.jpg -- that same data file, byte for byte (a fixture the reference's tests hold), and
   test_luma_256x256.u8 -- its luminance plane as libjpeg hands it out for JCS_GRAYSCALE (PIL ``draft('L')``), which is
   what IMREAD_GRAYSCALE asks libjpeg for: the bytes the reference's tests really see.  It differs from ``convert('L')``
   (RGB -> gray after colour conversion) at 202 pixels, by up to 10 levels.  The product's own JPEG frame source
   (include/canny_frames.h) must reproduce it exactly (tests/test_jpeg_gray.py).
2. oracle_stage_hashes.json -- SHA-256 of every stage output of the oracle (oracle/canny_oracle.c)
   on the fixture image and on small synthetic frames.  These are regression pins for the oracle
   itself (outputs of OUR restatement, not of the reference) so that an accidental change to the
   oracle or its build flags is caught on any machine.

reference_vectors.json is hand-transcribed data from tests/utils/test_utils.cpp and is not generated.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

REF_JPEG = "/root/reference/tests/test.jpg"


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    raw_path = os.path.join(HERE, "test_gray_256x256.u8")
    if os.path.exists(REF_JPEG):
        from PIL import Image
        g = np.asarray(Image.open(REF_JPEG).convert("L"), dtype=np.uint8)
        assert g.shape == (256, 256), g.shape
        g.tofile(raw_path)
        print("wrote", raw_path, sha(g)[:16])
        import shutil
        shutil.copyfile(REF_JPEG, os.path.join(HERE, "test.jpg"))
        im = Image.open(REF_JPEG)
        im.draft("L", im.size)
        y = np.asarray(im, dtype=np.uint8)
        assert im.mode == "L" and y.shape == (256, 256), (im.mode, y.shape)
        y.tofile(os.path.join(HERE, "test_luma_256x256.u8"))
        print("wrote test_luma_256x256.u8", sha(y)[:16], "differs from convert('L') at", int((y != g).sum()), "pixels")
    fixture = np.fromfile(raw_path, dtype=np.uint8).reshape(256, 256)

    luma = np.fromfile(os.path.join(HERE, "test_luma_256x256.u8"), dtype=np.uint8).reshape(256, 256)

    import oracle
    from canny_edge_amd.synth import synth_frame

    cases = {
        "fixture256_s0.5_50_150": (fixture, 0.5, 50, 150),
        "fixture256_s1.0_50_150": (fixture, 1.0, 50, 150),
        "jpegluma256_s1.0_50_150": (luma, 1.0, 50, 150),
        "synth_97x131_seed7_s1.4_50_150": (synth_frame(97, 131, 7), 1.4, 50, 150),
        "synth_240x320_seed42_s2.0_30_90": (synth_frame(240, 320, 42), 2.0, 30, 90),
        "synth_64x64_seed3_s0.5_10_50": (synth_frame(64, 64, 3), 0.5, 10, 50),
    }
    out = {}
    for name, (img, sigma, lo, hi) in cases.items():
        r = oracle.canny(img, sigma, lo, hi, stages=True)
        out[name] = {
            "shape": list(img.shape), "sigma": sigma, "min": lo, "max": hi,
            "input_sha256": sha(img),
            "smoothed_sha256": sha(r["smoothed"]), "magnitude_sha256": sha(r["magnitude"]),
            "angle_sha256": sha(r["angle"]), "nms_sha256": sha(r["nms"]), "edges_sha256": sha(r["edges"]),
            "nms_nonzero": int(np.count_nonzero(r["nms"])), "edge_pixels": int(np.count_nonzero(r["edges"])),
        }
        print(name, out[name]["nms_nonzero"], out[name]["edge_pixels"])
    with open(os.path.join(HERE, "oracle_stage_hashes.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
