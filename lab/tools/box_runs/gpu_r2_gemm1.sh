#!/bin/bash
# round 2, call 1: gemm8p correctness (smallest case first, under a short timeout: a barrier mismatch would hang), then
# the new full-width oracle tests, then the interleaved A/B against gemm256 and the vendor library
mkdir -p gpurun_out
( timeout 240 python -m pytest tests/test_gemm8p_gpu.py -x -q -k "test_gemm8p_bias and 256-256-64" 2>&1 | tail -5 ) > gpurun_out/r2_gemm8p_first.log
cat gpurun_out/r2_gemm8p_first.log
if grep -q "1 passed" gpurun_out/r2_gemm8p_first.log; then
  ( timeout 900 python -m pytest tests/test_gemm8p_gpu.py -q 2>&1 | tail -25 ) > gpurun_out/r2_gemm8p_tests.log
  cat gpurun_out/r2_gemm8p_tests.log
  ( timeout 600 python tools/gemm_ab.py 5 2>&1 | tail -20 ) > gpurun_out/r2_gemm_ab1.log
  cat gpurun_out/r2_gemm_ab1.log
fi
( timeout 900 python -m pytest tests/test_fullwidth_oracle_gpu.py tests/test_hip_ops.py -q 2>&1 | tail -25 ) > gpurun_out/r2_fullwidth_tests.log
cat gpurun_out/r2_fullwidth_tests.log
