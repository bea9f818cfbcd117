#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc counter_collection.csv files: per kernel, mean counter value per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
for path in sorted(glob.glob(f"{root}/*/*/*_counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:72]  # template arguments tell the variants apart
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", path.split("/")[-3])
    for k, cs in acc.items():
        if not k.startswith("canny"):
            continue
        print("  ", k.ljust(72), "  ".join(f"{c}={sum(v)/len(v):.4g}(n={len(v)})" for c, v in sorted(cs.items())))
