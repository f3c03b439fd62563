#!/bin/bash
# round 5, GPU call 3: gemm8p of this tree vs the round-4 build of the library (lab/r4lib), alternating processes on one box
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2; do
  echo "=== round-4 library, pass $rep"; timeout 200 tools/gemm_epi_ab_r4 3 0
  echo "=== this tree, pass $rep"; timeout 300 tools/gemm_epi_ab 3 1108,1000 1108,1001 1110,1001 1110,1007
done > gpurun_out/r05_gemm_vs_r4_library.log 2>&1
cat gpurun_out/r05_gemm_vs_r4_library.log
timeout 300 python tools/dbg_halo_first_frame.py > gpurun_out/dbg_halo.log 2>&1; tail -12 gpurun_out/dbg_halo.log
