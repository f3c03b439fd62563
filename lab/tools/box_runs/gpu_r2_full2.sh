#!/bin/bash
# round 2: drop-in / communicator / bench-workload checks, then the whole GPU suite and one C3 video
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_reference_caller_gpu.py tests/test_sp_gpu.py tests/test_capi.py -x -q 2>&1 | tail -25 ) > gpurun_out/r2_dropin_tests.log
cat gpurun_out/r2_dropin_tests.log
for wl in smoke_128p_17f c1_1024p_image c5_vae_768p_241f; do
  ( timeout 600 python bench.py --tiny-model --workload $wl --no-cpu-baseline 2>&1 | tail -2 | cut -c1-700 ) > gpurun_out/r2_bench_tiny_$wl.log
  cat gpurun_out/r2_bench_tiny_$wl.log
done
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > gpurun_out/r2_pytest_gpu2.log
cat gpurun_out/r2_pytest_gpu2.log
( timeout 900 python bench.py --steps 1 --warmup 0 2>gpurun_out/r2_bench_c3_b.err | tail -3 ) > gpurun_out/r2_bench_c3_b.log
cat gpurun_out/r2_bench_c3_b.log; tail -5 gpurun_out/r2_bench_c3_b.err
