#!/bin/bash
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_vae_gpu.py tests/test_pipeline_gpu.py tests/test_sp_gpu.py -x -q 2>&1 | tail -4 ) > gpurun_out/vae_tests.log
cat gpurun_out/vae_tests.log
timeout 900 python tools/vae_bench.py 4:1 4:4 4:8 2:8 > gpurun_out/vae_bench2.log 2>&1
grep "n_streams\|Error" gpurun_out/vae_bench2.log | cut -c1-250
