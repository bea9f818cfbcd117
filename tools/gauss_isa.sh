#!/bin/bash
# Device ISA of the Gaussian marching kernels for ONE half-window (default 5 = window 11), with register / spill counts
# per kernel and the instruction census of its loops: tools/gauss_isa.sh [C] -> /tmp/isa/gauss_c<C>.s
C=${1:-5}
mkdir -p /tmp/isa
cd "$(dirname "$0")/../canny_edge_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. \
    -fno-slp-vectorize -mllvm -amdgpu-sdwa-peephole=0 --cuda-device-only $GAUSS_FLAGS -DCANNY_GAUSS_ONLY_C=$C -S canny_gaussian_march.hip -o /tmp/isa/gauss_c$C.s 2>&1 | grep -v "hip-link"
grep -E "^\s+\.name:|\.vgpr_count|\.vgpr_spill_count|\.sgpr_spill_count|private_segment_fixed" /tmp/isa/gauss_c$C.s | grep -A4 "gauss_sym_kernel" | paste - - - - - | sed 's/ \+/ /g'
