#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -30 ) > gpurun_out/pytest_gpu.log
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 ) > gpurun_out/smoke.log
tail -n 22 gpurun_out/pytest_gpu.log | cut -c1-200; tail -n 2 gpurun_out/smoke.log
