#!/bin/bash
# round 4, call 11: rank-shape prediction of the final tree with the production single-GPU engine as the N = 1 / guidance rows
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python tools/rank_shape_bench.py --out gpurun_out/r4_rank_shape_final.json > gpurun_out/r4_rank_shape_final.log 2>&1
tail -16 gpurun_out/r4_rank_shape_final.log | cut -c1-260
