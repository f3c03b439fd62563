#!/bin/bash
# round 5, GPU call 4: gemm8p of this tree (static two-buffer ring again) vs the round-4 build, alternating processes
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2; do
  echo "=== round-4 library, pass $rep"; timeout 200 tools/gemm_epi_ab_r4 5 0
  echo "=== this tree, pass $rep"; timeout 300 tools/gemm_epi_ab 5 1000 1001 1003 1005 1007
done > gpurun_out/r05_gemm_vs_r4_library2.log 2>&1
cat gpurun_out/r05_gemm_vs_r4_library2.log
