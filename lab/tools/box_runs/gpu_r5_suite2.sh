#!/bin/bash
# the rest of the GPU tier after the grouped-forward test (same order as the full run), then the rank-shape harness A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1700 python -m pytest -q -m gpu -x --durations=15 tests/test_gemm_grouped_gpu.py tests/test_hip_ops.py tests/test_i2v_gpu.py tests/test_pipeline_gpu.py tests/test_qk_epilogue_gpu.py tests/test_reference_caller_gpu.py tests/test_sp_gpu.py tests/test_text_encoder_gpu.py tests/test_vae_gpu.py tests/test_video_io.py -s > gpurun_out/r05_pytest_gpu_rest.log 2>&1
tail -25 gpurun_out/r05_pytest_gpu_rest.log
grep -h "grouped vs two-stream" gpurun_out/r05_pytest_gpu_rest.log
GEMM_AB_SHAPES=8,9,10,11,12 timeout 200 tools/gemm_epi_ab 5 0 -8 > gpurun_out/r05_gemm_rank_shapes.log 2>&1
cat gpurun_out/r05_gemm_rank_shapes.log
