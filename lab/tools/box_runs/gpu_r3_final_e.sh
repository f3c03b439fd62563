#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 130 python bench.py > gpurun_out/r3_bench_c3_e.log 2> gpurun_out/r3_bench_c3_e.err; grep '^{' gpurun_out/r3_bench_c3_e.log | cut -c1-250
