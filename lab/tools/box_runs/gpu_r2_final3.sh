#!/bin/bash
# whole GPU suite + smoke + the C3 bench line on the final round-2 tree
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -m gpu -q --durations=5 2>&1 | grep -v "Warning\|^  " | tail -14 ) > gpurun_out/r2_pytest_gpu_final3.log
cat gpurun_out/r2_pytest_gpu_final3.log | cut -c1-200
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 ) > gpurun_out/r2_smoke_final3.log
cat gpurun_out/r2_smoke_final3.log
( timeout 900 python bench.py 2>gpurun_out/r2_bench_c3_final3.err | tail -1 ) > gpurun_out/r2_bench_c3_final3.log
cut -c1-700 gpurun_out/r2_bench_c3_final3.log; tail -2 gpurun_out/r2_bench_c3_final3.err
