#!/bin/bash
# A/B of prebuilt library variants inside ONE GPU session (box-to-box spread is +-3 %):
#   canny_edge_amd/libcanny_hip_<V>.so for V in $VARIANTS are swapped in turn under bench.py, $ROUNDS rounds.
set -e
VARIANTS=${VARIANTS:-"A B"}
ROUNDS=${ROUNDS:-2}
mkdir -p gpurun_out
cp canny_edge_amd/libcanny_hip.so canny_edge_amd/libcanny_hip_keep.so
for rnd in $(seq 1 $ROUNDS); do
  for v in $VARIANTS; do
    cp canny_edge_amd/libcanny_hip_$v.so canny_edge_amd/libcanny_hip.so
    timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-check > gpurun_out/ab_${v}_$rnd.log 2>&1
  done
done
cp canny_edge_amd/libcanny_hip_keep.so canny_edge_amd/libcanny_hip.so
