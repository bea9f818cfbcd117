#!/usr/bin/env python3
"""Builds profiles/traffic_sobel_nms.json (read by bench.py for `roofline.traffic`) from the FETCH_SIZE and
WRITE_SIZE passes of tools/pmc_passes.sh.

    python tools/make_traffic_json.py [pmc_dir=gpurun_out/pmc] [out=profiles/traffic_sobel_nms.json]

Corrections follow MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of a wide (16 B/lane) coalesced streaming read, so it is doubled; WRITE_SIZE is
exact for 16 B/lane streaming stores.  Only kernels whose accesses have that shape get a figure here (the two
Sobel+NMS kernels read 16 B/lane; the fused one writes single plane bytes, so its small write share -- 6 % of
its traffic -- is the raw counter and marked as such)."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

FRAMES, HEIGHT, WIDTH = 128, 2160, 3840  # bench.py's default workload (what tools/pmc_passes.sh runs)

# key in the JSON -> (substring of the rocprofv3 kernel name, algorithmic bytes per pixel, frames per launch, note)
# (with the option overlap_hysteresis=1 canny() would launch the fused kernel once per half of the batch)
KERNELS = {
    # what canny() runs since round 3: the f32 fused kernel on the u8 smoothed plane (8 B/lane reads)
    "sobel_nms_classify_u8in": ("sobel_nms_march_kernel<true, 4, true, true, true>", 3.25, FRAMES,
                                "8 B/lane reads (FETCH_SIZE doubled: the counter tallies 64 B per 128 B request whatever "
                                "the lane width -- the u8 Gaussian's 4 B/lane reads show the same half); 16 B/lane "
                                "edge-map writes (exact) plus the plane bytes (raw WRITE_SIZE)"),
    # the same kernel on the s16 plane (canny() with smoothed_u8 = 0; bench.py times it after the region)
    "sobel_nms_classify": ("sobel_nms_march_kernel<true, 4, true, false, true>", 4.25, FRAMES,
                           "16 B/lane reads (FETCH_SIZE doubled); 16 B/lane edge-map writes (exact) plus the plane "
                           "bytes (raw WRITE_SIZE, 6 % of the writes)"),
    # the stage-API kernel SURVEY 8(d) prices (packed-i16 arithmetic)
    "sobel_nms": ("sobel_nms_march_kernel<false, 4, false, false, false>", 4.0, FRAMES,
                  "16 B/lane reads (FETCH_SIZE doubled) and writes (exact)"),
    "gaussian_u8out": ("gauss_sym_kernel<5, true, true>", 2.0, FRAMES,
                       "4 B/lane reads (FETCH_SIZE doubled, see above) and 4 B/lane writes (raw WRITE_SIZE)"),
}


def mean_per_dispatch(pmc_dir, sub, counter):
    acc = collections.defaultdict(list)
    for path in glob.glob(f"{pmc_dir}/{sub}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    pmc_dir = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
    out = sys.argv[2] if len(sys.argv) > 2 else "profiles/traffic_sobel_nms.json"
    fetch = mean_per_dispatch(pmc_dir, "fetch", "FETCH_SIZE")
    write = mean_per_dispatch(pmc_dir, "write", "WRITE_SIZE")
    kernels = {}
    for key, (needle, bpp, frames_per_launch, note) in KERNELS.items():
        px = frames_per_launch * HEIGHT * WIDTH
        f = [(v, n) for k, (v, n) in fetch.items() if needle in k]
        w = [(v, n) for k, (v, n) in write.items() if needle in k]
        if not f or not w:
            continue
        fetch_b = 2.0 * f[0][0] * 1024.0
        write_b = w[0][0] * 1024.0
        alg = bpp * px
        kernels[key] = {
            "kernel": needle, "frames_per_launch": frames_per_launch, "dispatches_averaged": [f[0][1], w[0][1]],
            "fetch_size_kib_raw": round(f[0][0], 1), "write_size_kib_raw": round(w[0][0], 1),
            "fetch_bytes_corrected": int(fetch_b), "write_bytes": int(write_b),
            "hbm_bytes_per_launch": int(fetch_b + write_b),
            "algorithmic_bytes_per_launch": int(alg), "traffic_over_algorithmic": round((fetch_b + write_b) / alg, 3),
            "note": note,
        }
    doc = {
        "_comment": "HBM traffic per launch in bench.py's default workload from rocprofv3 PMC passes "
                    "(tools/pmc_passes.sh, one counter per pass, --kernel-trace only); units and the gfx950 "
                    "FETCH_SIZE x2 correction per MI355X_MICROARCH.md (HBM).  Raw per-dispatch rows: "
                    "profiles/rNN_pmc/fetch_counter_collection.csv, write_counter_collection.csv.",
        "frames": FRAMES, "height": HEIGHT, "width": WIDTH, "kernels": kernels,
        # bench.py reports `traffic` only while the kernel source is the one these counters were collected on
        "kernel_source_sha256": hashlib.sha256(open(os.path.join(
            os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "canny_edge_amd", "csrc",
            "canny_sobel_nms_march.hip"), "rb").read()).hexdigest(),
    }
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
