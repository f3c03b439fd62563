#!/bin/bash
# round 4, call 14: the guidance-parallel path after sp.pair_gather (python-only change): its GPU tests + the self-launch test
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_sp_gpu.py tests/test_bench_selflaunch_gpu.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r4_call14_pytest.log
cat gpurun_out/r4_call14_pytest.log
