#!/bin/bash
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_vae_gpu.py tests/test_pipeline_gpu.py -x -q 2>&1 | tail -4 ) > gpurun_out/vae_tests.log
( timeout 900 python tools/vae_bench.py 1 2 4 8 2>&1 | tail -6 ) > gpurun_out/vae_bench.log
cat gpurun_out/vae_tests.log gpurun_out/vae_bench.log
