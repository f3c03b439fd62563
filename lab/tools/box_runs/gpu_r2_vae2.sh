#!/bin/bash
mkdir -p gpurun_out
( for i in 1 2; do
    PF_BENCH_LIB=libpyflow_hip_prev.so timeout 300 python tools/vae_bench.py 4:4 2>&1 | tail -1 | sed 's/^/prev: /'
    timeout 300 python tools/vae_bench.py 4:4 2>&1 | tail -1 | sed 's/^/new:  /'
  done ) > gpurun_out/r2_vae_tileorder_ab.log
cat gpurun_out/r2_vae_tileorder_ab.log
