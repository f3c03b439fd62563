#!/bin/bash
# round 4, call 21: the -m gpu suite of the end-of-round tree without the two multi-process files (those ran in call 14; the whole
# suite is 12 minutes, more than the budget left)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 560 python -m pytest tests -m gpu -q --ignore=tests/test_sp_gpu.py --ignore=tests/test_bench_selflaunch_gpu.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo\|^$" | tail -12 ) > gpurun_out/r4_pytest_gpu_end_of_round.log
cat gpurun_out/r4_pytest_gpu_end_of_round.log | cut -c1-200
