#!/bin/bash
# round 4, call 18: the GEMM tests + smoke after the (default-off) desynchronised-start hook went into gemm8p
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 800 python -m pytest tests/test_gemm8p_gpu.py tests/test_qk_epilogue_gpu.py tests/test_gemm256_gpu.py tests/test_hip_ops.py -x -q -m gpu 2>&1 | tail -5
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > gpurun_out/r4_call18.log 2>&1
cat gpurun_out/r4_call18.log | grep -v amdgpu.ids
