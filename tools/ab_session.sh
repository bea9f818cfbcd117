#!/bin/bash
# One A/B SESSION (= one GPU box): the Sobel+NMS arithmetic variants with and without the u8 smoothed plane, interleaved
# rounds in separate processes (tools/ab_stage_times.py).  Run >= 3 sessions (boxes differ by up to 14 %).
#   tools/ab_session.sh <tag> [rounds]
TAG=${1:-s1}
ROUNDS=${2:-3}
mkdir -p gpurun_out/r3
timeout -k 10 850 python tools/ab_stage_times.py --rounds "$ROUNDS" \
    flt: pk16:tune_sobel_variant=1 flt_u8:smoothed_u8=1 pk16_u8:tune_sobel_variant=1,smoothed_u8=1 \
    > gpurun_out/r3/ab_session_$TAG.txt 2>&1
tail -5 gpurun_out/r3/ab_session_$TAG.txt
