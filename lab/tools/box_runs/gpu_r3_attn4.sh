#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest -x -q -s -m gpu tests/test_attention_w64_gpu.py tests/test_fullsize_gpu.py tests/test_hip_ops.py tests/test_blocks_gpu.py tests/test_flux_forward_gpu.py tests/test_cmdlist_gpu.py tests/test_sp_gpu.py > gpurun_out/r3_attn_tests4.log 2>&1
echo "exit $?" >> gpurun_out/r3_attn_tests4.log
grep -v "^$" gpurun_out/r3_attn_tests4.log | tail -30

