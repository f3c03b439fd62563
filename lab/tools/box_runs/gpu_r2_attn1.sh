#!/bin/bash
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_hip_ops.py tests/test_fullsize_gpu.py tests/test_blocks_gpu.py -q 2>&1 | tail -12 ) > gpurun_out/r2_attn_tests1.log
cat gpurun_out/r2_attn_tests1.log
( timeout 300 python tools/attn_bench.py 2>&1 | tail -6 ) > gpurun_out/r2_attn_bench1.log
cat gpurun_out/r2_attn_bench1.log
