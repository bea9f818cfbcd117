#!/bin/bash
# Produces the evidence files of a round on the GPU box (run from the repo root through gpurun):
#   gpurun_out/bench.json.log      the JSON line of an un-profiled default bench.py run
#   gpurun_out/prof/               rocprofv3 --kernel-trace --stats of the same command (no CPU baseline)
#   gpurun_out/pmc/                PMC passes (tools/pmc_passes.sh) + summary.txt + traffic.json
# Copy what should be judged into profiles/ afterwards (gpurun_out/ is scratch).
set -u
ROOT=$(pwd)
mkdir -p gpurun_out
timeout -k 10 420 python3 bench.py --steps 20 --warmup 3 > gpurun_out/bench.json.log 2> gpurun_out/bench.err || { echo "bench failed"; tail -5 gpurun_out/bench.err; exit 1; }
tail -c 600 gpurun_out/bench.json.log; echo
rm -rf gpurun_out/prof
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof" -- python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --h2h-frames 0 > "$ROOT/gpurun_out/prof.log" 2>&1) || { echo "rocprofv3 stats failed"; tail -5 gpurun_out/prof.log; exit 1; }
echo "stats done"
rm -rf gpurun_out/pmc
timeout -k 10 600 bash tools/pmc_passes.sh gpurun_out/pmc > gpurun_out/pmc_passes.log 2>&1 || { echo "pmc failed"; tail -5 gpurun_out/pmc_passes.log; exit 1; }
python3 tools/pmc_summary.py gpurun_out/pmc > gpurun_out/pmc/summary.txt
python3 tools/make_traffic_json.py gpurun_out/pmc gpurun_out/pmc/traffic.json > /dev/null
grep -c . gpurun_out/pmc/summary.txt
