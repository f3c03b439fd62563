#!/bin/bash
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gemm256_gpu.py -x -q 2>&1 | tail -5 ) > gpurun_out/gemm256_tests.log
( timeout 600 python tools/microbench.py gemm 2>&1 | tail -12 ) > gpurun_out/gemm256_bench5.log
cat gpurun_out/gemm256_tests.log gpurun_out/gemm256_bench5.log
