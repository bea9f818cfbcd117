#!/bin/bash
# PMC passes over tools/probe_sobel_pmc.py (both arithmetic variants of the marching Sobel+NMS kernel).
set -u
OUT=${1:-gpurun_out/pmc_sobel}
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT \
    --kernel-trace --output-format csv -d "$ROOT/$OUT/sq" -- python3 "$ROOT/tools/probe_sobel_pmc.py" > "$ROOT/$OUT/sq.log" 2>&1
echo "pass sq rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES \
    --kernel-trace --output-format csv -d "$ROOT/$OUT/sq2" -- python3 "$ROOT/tools/probe_sobel_pmc.py" > "$ROOT/$OUT/sq2.log" 2>&1
echo "pass sq2 rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$ROOT/$OUT/fetch" -- python3 "$ROOT/tools/probe_sobel_pmc.py" > "$ROOT/$OUT/fetch.log" 2>&1
echo "pass fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$ROOT/$OUT/write" -- python3 "$ROOT/tools/probe_sobel_pmc.py" > "$ROOT/$OUT/write.log" 2>&1
echo "pass write rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats" -- python3 "$ROOT/tools/probe_sobel_pmc.py" > "$ROOT/$OUT/stats.log" 2>&1
echo "pass stats rc=$?"
