#!/bin/bash
# config C1 (1024x1024 image, one pyramid stage, 20 steps) at full size on the GPU: the counterpart of profiles/r02_c1_cpu_reference.log
mkdir -p gpurun_out
( timeout 400 python bench.py --workload c1_1024p_image --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 ) > gpurun_out/r2_bench_c1.log
cut -c1-900 gpurun_out/r2_bench_c1.log
