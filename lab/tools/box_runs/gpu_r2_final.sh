#!/bin/bash
# round 2, final tree: whole GPU suite + smoke + the C3 bench line (default options)
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -m gpu -q --durations=6 2>&1 | tail -24 ) > gpurun_out/r2_pytest_gpu_final.log
cat gpurun_out/r2_pytest_gpu_final.log | cut -c1-200
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 ) > gpurun_out/r2_smoke_final.log
cat gpurun_out/r2_smoke_final.log
( timeout 900 python bench.py 2>gpurun_out/r2_bench_c3_final2.err | tail -2 ) > gpurun_out/r2_bench_c3_final2.log
cut -c1-2500 gpurun_out/r2_bench_c3_final2.log; tail -3 gpurun_out/r2_bench_c3_final2.err
