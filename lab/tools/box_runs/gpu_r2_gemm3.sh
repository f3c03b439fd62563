#!/bin/bash
# gemm8p with the quadrant-pipelined epilogue: parity tests first, then the A/B against gemm256 and the vendor library
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gemm8p_gpu.py tests/test_gemm256_gpu.py -q -x 2>&1 | tail -6 ) > gpurun_out/r2_gemm_tests3.log
cat gpurun_out/r2_gemm_tests3.log
( timeout 600 python tools/gemm_ab.py 3 2>&1 | tail -14 ) > gpurun_out/r2_gemm_ab3.log
cat gpurun_out/r2_gemm_ab3.log
