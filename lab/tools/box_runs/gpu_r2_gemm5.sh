#!/bin/bash
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gemm8p_gpu.py tests/test_gemm256_gpu.py -q -x 2>&1 | tail -4 ) > gpurun_out/r2_gemm_tests5.log
cat gpurun_out/r2_gemm_tests5.log
bash tools/gpu_r2_gemm4.sh "libpyflow_hip_old.so libpyflow_hip.so" 1,2,4,6,10
