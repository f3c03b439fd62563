#!/bin/bash
# round 5, the budget's last two minutes: the N > 1 bench path (self-test of the communicator restructured this round) on two ranks of one GPU
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 125 python -m pytest -q -m gpu -p no:cacheprovider "tests/test_bench_selflaunch_gpu.py::test_self_launch_sequence_parallel_two_ranks_and_native_communicator_dry_run" "tests/test_bench_selflaunch_gpu.py::test_self_launch_default_two_ranks_is_guidance_parallel" > gpurun_out/r05_tail2_tests.log 2>&1
tail -4 gpurun_out/r05_tail2_tests.log
