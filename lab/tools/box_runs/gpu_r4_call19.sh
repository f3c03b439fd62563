#!/bin/bash
# round 4, call 19: the C3 bench line of the end-of-round tree (library = call 18's; python = the guidance pair_gather refactor)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( time timeout 700 python bench.py --steps 2 --warmup 1 ) > gpurun_out/r4_bench_c3_end_of_round.log 2>&1
tail -n 5 gpurun_out/r4_bench_c3_end_of_round.log | cut -c1-700
