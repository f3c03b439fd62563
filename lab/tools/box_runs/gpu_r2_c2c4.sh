#!/bin/bash
# configs C2 (384p, 121 frames) and C4 (768p image-to-video, SD3 MMDiT) on the final round-2 tree
mkdir -p gpurun_out
: > gpurun_out/r2_bench_c2_c4.log
for wl in c2_384p_121f c4_i2v_768p_121f; do
  ( timeout 110 python bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-1200 ) >> gpurun_out/r2_bench_c2_c4.log
done
cut -c1-330 gpurun_out/r2_bench_c2_c4.log
