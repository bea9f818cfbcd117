#!/usr/bin/env python3
"""Interleaved A/B of library builds and/or option sets in ONE GPU session (boxes differ by +-4 %, runs by +-3 %):

    python tools/ab_stage_times.py --rounds 3 name1:lib=<path.so>,opt=val,... name2:...

Every variant runs tools/stage_times.py in its own process, round-robin; prints the median per stage."""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    rounds, frames = 3, 128
    variants = []
    i = 0
    while i < len(args):
        if args[i] == "--rounds":
            rounds = int(args[i + 1]); i += 2
        elif args[i] == "--frames":
            frames = int(args[i + 1]); i += 2
        else:
            name, _, spec = args[i].partition(":")
            lib, opts = None, []
            for kv in filter(None, spec.split(",")):
                k, v = kv.split("=")
                if k == "lib":
                    lib = v
                else:
                    opts.append(f"{k}={v}")
            variants.append((name, lib, opts)); i += 1
    res = {name: [] for name, _, _ in variants}
    for r in range(rounds):
        for name, lib, opts in variants:
            env = dict(os.environ)
            if lib:
                env["CANNY_HIP_LIB"] = os.path.join(ROOT, lib)
            cmd = [sys.executable, os.path.join(ROOT, "tools", "stage_times.py"), "--frames", str(frames)]
            if r == 0:
                cmd.append("--check")
            for o in opts:
                cmd += ["--opt", o]
            p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
            if p.returncode != 0:
                print(f"[{name}] FAILED: {p.stderr[-400:]}", flush=True)
                continue
            d = json.loads(p.stdout.strip().splitlines()[-1])
            res[name].append(d)
            print(f"[round {r}] {name}: {p.stdout.strip().splitlines()[-1]}", flush=True)
    print("\n==== medians (ms per 128-frame step) ====")
    for name, runs in res.items():
        if not runs:
            continue
        med = lambda f: round(statistics.median([f(x) for x in runs if f(x) is not None]), 4)
        g = med(lambda x: x["canny_stages_ms"].get("gaussian"))
        s = med(lambda x: x["canny_stages_ms"].get("sobel_nms"))
        pr = med(lambda x: x["canny_stages_ms"].get("hyst_propagate"))
        wall = med(lambda x: x["canny_wall_ms"])
        s16 = med(lambda x: x.get("sobel_nms_s16_ms"))
        extra = ""
        if all("sobel_nms_u8in_ms" in x for x in runs):
            extra = f"  sobel_nms(u8 in) {med(lambda x: x['sobel_nms_u8in_ms'])}  gaussian(u8 out) {med(lambda x: x['gaussian_u8_ms'])}"
        extra += f"  1 frame: latency {med(lambda x: x.get('single_frame_latency_ms'))} stream {med(lambda x: x.get('single_frame_stream_ms'))}"
        par = sorted({x["edges_sha256"] for x in runs if "edges_sha256" in x})  # equal across variants = same edge map
        print(f"{name:28s} canny wall {wall}  gaussian {g}  sobel+nms+classify {s}  propagate {pr}  sobel_nms(s16) {s16}{extra}  edges_sha256={par}")


if __name__ == "__main__":
    main()
