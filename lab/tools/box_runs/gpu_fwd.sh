#!/bin/bash
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_flux_forward_gpu.py tests/test_pipeline_gpu.py -x -q 2>&1 | tail -4 ) > gpurun_out/fwd_tests.log
( timeout 900 python tools/forward_bench.py 2>&1 | tail -6 ) > gpurun_out/fwd_bench.log
cat gpurun_out/fwd_tests.log gpurun_out/fwd_bench.log
