#!/bin/bash
# SQ counter passes over tools/probe_gauss_variants_pmc.py (run on the GPU box from the repo root); PMC passes carry
# --kernel-trace only.
set -u
OUT=${1:-gpurun_out/pmc_gauss}
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT \
    --kernel-trace --output-format csv -d "$ROOT/$OUT/sq" -- python3 "$ROOT/tools/probe_gauss_variants_pmc.py" > "$ROOT/$OUT/sq.log" 2>&1
echo "pass sq rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
    --kernel-trace --output-format csv -d "$ROOT/$OUT/sq2" -- python3 "$ROOT/tools/probe_gauss_variants_pmc.py" > "$ROOT/$OUT/sq2.log" 2>&1
echo "pass sq2 rc=$?"
python3 "$ROOT/tools/pmc_summary.py" "$ROOT/$OUT" > "$ROOT/$OUT/summary.txt" 2>&1
cat "$ROOT/$OUT/summary.txt"
