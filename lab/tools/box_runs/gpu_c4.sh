#!/bin/bash
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_flux_forward_gpu.py tests/test_i2v_gpu.py -x -q 2>&1 | tail -4 ) > gpurun_out/c4_tests.log
( time timeout 900 python bench.py --workload c4_i2v_768p_121f --no-cpu-baseline 2>&1 | tail -3 ) > gpurun_out/bench_c4.log 2>&1
cat gpurun_out/c4_tests.log; cut -c1-900 gpurun_out/bench_c4.log
