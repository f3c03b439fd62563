#!/bin/bash
# round 4, call 4: mid-size GEMM split, attention KV split, guidance-parallel engine, QK epilogue (rest of the files), SP suite,
# then the rank-shape prediction with the fixes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_attention_w64_gpu.py tests/test_gemm8p_gpu.py tests/test_qk_epilogue_gpu.py tests/test_hip_ops.py tests/test_flux_forward_gpu.py tests/test_cmdlist_gpu.py tests/test_blocks_gpu.py tests/test_pipeline_gpu.py tests/test_sp_gpu.py tests/test_bench_selflaunch_gpu.py -m gpu -q --durations=8 2>&1 | grep -v "^$" | tail -60 ) > gpurun_out/r4_call4_pytest.log
cat gpurun_out/r4_call4_pytest.log | cut -c1-250
( timeout 600 python tools/rank_shape_bench.py --out gpurun_out/r4_rank_shape_fixed.json 2>&1 | tail -170 ) > gpurun_out/r4_rank_shape_fixed.log
tail -28 gpurun_out/r4_rank_shape_fixed.log | cut -c1-260
