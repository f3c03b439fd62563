#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_text_encoder_gpu.py tests/test_hip_ops.py tests/test_gemm256_gpu.py -x -q > gpurun_out/text_tests.log 2>&1
tail -n 25 gpurun_out/text_tests.log | cut -c1-300
timeout 600 python tools/text_bench.py 10 > gpurun_out/text_bench.log 2>&1
tail -n 8 gpurun_out/text_bench.log | cut -c1-600
