#!/bin/bash
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_hip_ops.py tests/test_flux_forward_gpu.py -x -q 2>&1 | tail -5 ) > gpurun_out/attn_tests.log
( timeout 600 python tools/microbench.py attn 2>&1 | tail -12 ) > gpurun_out/attn_bench.log
cat gpurun_out/attn_tests.log gpurun_out/attn_bench.log
