#!/bin/bash
# round 4, call 17: pf_gemm_set_policy(9) (desynchronised start of gemm8p's workgroups) same-box A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 300 python tools/gemm_stagger_ab.py ) > gpurun_out/r4_gemm_stagger_ab.log 2>&1
cat gpurun_out/r4_gemm_stagger_ab.log | grep -v amdgpu.ids | cut -c1-220
