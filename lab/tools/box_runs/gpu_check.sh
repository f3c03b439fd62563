#!/bin/bash
# One gpurun call: GPU parity tests, smoke, a short C2 bench and the full C3 bench. Outputs under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/pytest_gpu.log
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 ) > gpurun_out/smoke.log
( time timeout 600 python bench.py --workload c2_384p_121f --no-cpu-baseline 2>&1 | tail -5 ) > gpurun_out/bench_c2.log 2>&1
( time timeout 1200 python bench.py 2>&1 | tail -5 ) > gpurun_out/bench_c3.log 2>&1
for f in pytest_gpu smoke bench_c2 bench_c3; do tail -n 3 gpurun_out/$f.log | cut -c1-1500; done
