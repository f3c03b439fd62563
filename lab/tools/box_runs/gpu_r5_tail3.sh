#!/bin/bash
# round 5, the budget's last minute: the C-ABI communicator on one rank + the replicas launch
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 80 python -m pytest -q -m gpu -p no:cacheprovider "tests/test_sp_gpu.py::test_native_communicator_on_one_rank" "tests/test_sp_gpu.py::test_rccl_api_on_one_rank" "tests/test_bench_selflaunch_gpu.py::test_self_launch_replicas_two_ranks" > gpurun_out/r05_tail3_tests.log 2>&1
tail -4 gpurun_out/r05_tail3_tests.log
