#!/bin/bash
# Samples rocm-smi clocks and power while bench.py keeps the device busy (read-only; informs the DVFS notes in DESIGN.md)
mkdir -p gpurun_out
(timeout -k 10 120 python bench.py --steps 3000 --warmup 3 --no-cpu-baseline --no-check --spinup-seconds 0 > gpurun_out/clk_bench.log 2>&1) &
BP=$!
sleep 4   # import + setup
for i in $(seq 1 10); do
  echo "--- t=$i"; rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (junction|edge)" | head -8
  sleep 0.7
done
wait $BP
tail -c 400 gpurun_out/clk_bench.log
