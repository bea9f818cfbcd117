#!/bin/bash
# Rebuilds the Gaussian TU on the GPU box with small source variations and times each (scratch tool).
set -u
F=canny_edge_amd/csrc/canny_gaussian_march.hip
cp $F /tmp/gm_orig.hip
run() { make -C canny_edge_amd/csrc -j8 2>&1 | grep -E " error" ; python tools/tune_stages.py --rounds 3 2>&1 | grep -E "gaussian"; }
echo "== A: as committed (volatile window reads, fence only)"; run
echo "== B: non-volatile window reads, fence only"
sed -i 's/reinterpret_cast<const volatile f32x4 \*>/reinterpret_cast<const f32x4 *>/' $F; run
echo "== C: non-volatile + wave_barrier"
sed -i 's/__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }/__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }/' $F; run
echo "== D: volatile + wave_barrier"
sed -i 's/reinterpret_cast<const f32x4 \*>/reinterpret_cast<const volatile f32x4 *>/' $F; run
cp /tmp/gm_orig.hip $F
