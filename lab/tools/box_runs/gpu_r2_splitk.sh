#!/bin/bash
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_hip_ops.py tests/test_fullwidth_oracle_gpu.py tests/test_blocks_gpu.py tests/test_cmdlist_gpu.py tests/test_text_encoder_gpu.py tests/test_flux_forward_gpu.py -x -q 2>&1 | tail -12 ) > gpurun_out/r2_splitk_tests.log
cat gpurun_out/r2_splitk_tests.log
( timeout 600 python tools/host_overhead.py 2>&1 | grep "forward" ) > gpurun_out/r2_host_overhead2.log
cat gpurun_out/r2_host_overhead2.log
( timeout 300 python tools/text_bench.py 2>&1 | tail -8 ) > gpurun_out/r2_text_bench.log
cat gpurun_out/r2_text_bench.log
