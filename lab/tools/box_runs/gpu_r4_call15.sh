#!/bin/bash
# round 4, call 15: lab/gemm4w_lab (four-wave hand-placed GEMM main loop) next to pf_gemm_bf16 on the DiT's shapes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 300 lab/gemm4w_lab > gpurun_out/r4_gemm4w_lab.log 2>&1
cat gpurun_out/r4_gemm4w_lab.log | cut -c1-230
