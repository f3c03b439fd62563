#!/bin/bash
# round 4, call 16: gemm4w persistent form: small problems (1, 2, 4 tiles per workgroup), then the DiT's shapes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( echo "== 1024 1024 512 grid 8 (2 tiles / WG)"; timeout 60 lab/gemm4w_lab 1024 1024 512 8 | grep "persistent (\|one launch per tile ("
  echo "== 2048 1024 512 grid 8 (4 tiles / WG)"; timeout 60 lab/gemm4w_lab 2048 1024 512 8 | grep "persistent (\|one launch per tile ("
  timeout 200 lab/gemm4w_lab | grep -v "^   \[\|main loop" ) > gpurun_out/r4_gemm4w_lab.log 2>&1
cut -c1-330 gpurun_out/r4_gemm4w_lab.log
