#!/bin/bash
# (the epilogue-diagnostic build this script also ran -- stores / whole epilogue compiled out -- was a temporary kernel variant; its output is profiles/r02_gemm8p_epilogue_diag.log)
mkdir -p gpurun_out
( timeout 600 python tools/gemm_ab.py 3 2>&1 | tail -14 ) > gpurun_out/r2_gemm_ab2.log
cat gpurun_out/r2_gemm_ab2.log
( timeout 600 python -m pytest tests/test_gemm8p_gpu.py tests/test_gemm256_gpu.py -q -x 2>&1 | tail -4 ) > gpurun_out/r2_gemm_tests2.log
cat gpurun_out/r2_gemm_tests2.log
