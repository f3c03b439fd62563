#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest -x -q -s -m gpu tests/test_sp_gpu.py "tests/test_fullsize_gpu.py::test_full_size_forward_invariants" tests/test_bench_selflaunch_gpu.py > gpurun_out/r3_sp_tests1.log 2>&1
echo "exit $?" >> gpurun_out/r3_sp_tests1.log
grep -v "^$" gpurun_out/r3_sp_tests1.log | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -30
