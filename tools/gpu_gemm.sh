#!/bin/bash
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gemm256_gpu.py tests/test_hip_ops.py tests/test_flux_forward_gpu.py -x -q 2>&1 | tail -5 ) > gpurun_out/gemm256_tests.log
( timeout 600 python tools/forward_bench.py 2>&1 | tail -5 ) > gpurun_out/fwd_bench2.log
cat gpurun_out/gemm256_tests.log gpurun_out/fwd_bench2.log
