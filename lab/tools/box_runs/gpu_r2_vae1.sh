#!/bin/bash
# round 2: VAE after the batched cache shifts / stats arena: tests, C5 decode bench, kernel stats of the decode
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$(pwd)
( timeout 900 python -m pytest tests/test_vae_gpu.py tests/test_i2v_gpu.py tests/test_fullwidth_oracle_gpu.py tests/test_pipeline_gpu.py -x -q 2>&1 | tail -6 ) > gpurun_out/r2_vae_tests1.log
cat gpurun_out/r2_vae_tests1.log
( timeout 600 python bench.py --workload c5_vae_768p_241f --steps 1 --warmup 1 2>gpurun_out/r2_bench_c5.err | tail -2 ) > gpurun_out/r2_bench_c5.log
cat gpurun_out/r2_bench_c5.log | cut -c1-2500; tail -3 gpurun_out/r2_bench_c5.err
cd /tmp
( timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_c5 -o c5 --output-format csv -- python $REPO/bench.py --workload c5_vae_768p_241f --steps 1 --warmup 0 ) > $REPO/gpurun_out/r2_rocprof_c5.log 2>&1
find /tmp/prof_c5 -name '*kernel_stats*' -exec cp {} $REPO/gpurun_out/r2_c5_kernel_stats.csv \;
head -30 $REPO/gpurun_out/r2_c5_kernel_stats.csv | cut -c1-200
