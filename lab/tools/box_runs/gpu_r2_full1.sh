#!/bin/bash
# round 2: full GPU test suite with gemm8p on by default + one C3 video
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r2_pytest_gpu1.log
cat gpurun_out/r2_pytest_gpu1.log
( timeout 900 python bench.py --steps 1 --warmup 0 2>gpurun_out/r2_bench_c3_a.err | tail -3 ) > gpurun_out/r2_bench_c3_a.log
cat gpurun_out/r2_bench_c3_a.log; tail -5 gpurun_out/r2_bench_c3_a.err
