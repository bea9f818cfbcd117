#!/bin/bash
# Collects rocprofv3 PMC passes for bench.py (run on the GPU box from the repo root):
#   pass A: SQ issue/wait/LDS counters   pass B: FETCH_SIZE   pass C: WRITE_SIZE
# Each pass is its own run with --kernel-trace only (never combined with sys/hip tracing).
set -u
OUT=${1:-gpurun_out/pmc}
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-check --spinup-seconds 0 --h2h-frames 0"
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT \
    --kernel-trace --output-format csv -d "$ROOT/$OUT/sq" -- python3 "$ROOT/bench.py" $ARGS > "$ROOT/$OUT/sq.log" 2>&1
echo "pass sq rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES \
    --kernel-trace --output-format csv -d "$ROOT/$OUT/sq2" -- python3 "$ROOT/bench.py" $ARGS > "$ROOT/$OUT/sq2.log" 2>&1
echo "pass sq2 rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$ROOT/$OUT/fetch" -- python3 "$ROOT/bench.py" $ARGS > "$ROOT/$OUT/fetch.log" 2>&1
echo "pass fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$ROOT/$OUT/write" -- python3 "$ROOT/bench.py" $ARGS > "$ROOT/$OUT/write.log" 2>&1
echo "pass write rc=$?"
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$ROOT/$OUT/tcc" -- python3 "$ROOT/bench.py" $ARGS > "$ROOT/$OUT/tcc.log" 2>&1
echo "pass tcc rc=$?"
find "$ROOT/$OUT" -name "*.csv" | head -30
