#!/bin/bash
# round 5, GPU call 5: GEMM parity tests on the final epilogue form + one C3 video
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest -x -q -m gpu -s tests/test_gemm8p_gpu.py tests/test_qk_epilogue_gpu.py \
   "tests/test_fullsize_gpu.py::test_full_size_gemm_vs_library" \
   "tests/test_fullsize_gpu.py::test_full_size_forward_vs_oracle_one_block_of_each_kind" \
   tests/test_convhalo_gpu.py tests/test_cmdlist_gpu.py \
   > gpurun_out/r05_gemm_epilogue_tests.log 2>&1
tail -4 gpurun_out/r05_gemm_epilogue_tests.log
timeout 600 python bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r05_bench_c3_epilogue.log 2>&1
tail -c 3000 gpurun_out/r05_bench_c3_epilogue.log
