#!/bin/bash
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_cmdlist_gpu.py tests/test_blocks_gpu.py tests/test_fullwidth_oracle_gpu.py -x -q 2>&1 | tail -8 ) > gpurun_out/r2_cmdlist_tests2.log
cat gpurun_out/r2_cmdlist_tests2.log
( timeout 900 python bench.py --steps 1 --warmup 0 2>gpurun_out/r2_bench_c3_graph.err | tail -2 ) > gpurun_out/r2_bench_c3_graph.log
cut -c1-400 gpurun_out/r2_bench_c3_graph.log; tail -3 gpurun_out/r2_bench_c3_graph.err
