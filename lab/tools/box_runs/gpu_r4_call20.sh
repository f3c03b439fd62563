#!/bin/bash
# round 4, call 20: gemm4w with a 256 x 192 tile on the N = 1920 / 5760 projections (no half-empty column tile; 10 / 30 tiles wide)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( for s in "30976 1920 1920" "30976 1920 7680" "30976 5760 1920" "30976 7680 1920"; do timeout 100 lab/gemm4w_lab $s | grep -v "^   \[\|main loop\|staggered"; done ) > gpurun_out/r4_gemm4w_192.log 2>&1
cut -c1-260 gpurun_out/r4_gemm4w_192.log
