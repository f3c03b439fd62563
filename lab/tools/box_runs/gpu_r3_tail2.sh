#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm8p_gpu.py -x -q -s > gpurun_out/r3_tail_tests3.log 2>&1; rc=$?; echo "gemm8p pytest rc=$rc"; tail -5 gpurun_out/r3_tail_tests3.log
[ $rc -ne 0 ] && exit 0
timeout 900 python tools/gemm_tail_ab.py 3 > gpurun_out/r3_gemm_tail_ab4.log 2>&1; tail -14 gpurun_out/r3_gemm_tail_ab4.log
