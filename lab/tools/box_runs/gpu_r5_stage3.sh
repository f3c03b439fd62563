#!/bin/bash
# round 5: the K split of skinny problems capped at 256 workgroups + the three-stage ring: tests of everything that runs the 128 x 128 kernel
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest -x -q -m gpu tests/test_hip_ops.py tests/test_text_encoder_gpu.py tests/test_flux_forward_gpu.py tests/test_vae_gpu.py tests/test_cmdlist_gpu.py tests/test_pipeline_gpu.py tests/test_gemm_grouped_gpu.py tests/test_blocks_gpu.py tests/test_fullwidth_oracle_gpu.py "tests/test_sp_gpu.py::test_sp_multi_process_exchange" "tests/test_sp_gpu.py::test_sp_engine_single_rank_matches_plain_engine" > gpurun_out/r05_gemm128_tests.log 2>&1
tail -3 gpurun_out/r05_gemm128_tests.log
GEMM_AB_SHAPES=13,14,15 timeout 100 tools/gemm_epi_ab 5 1205,1202 0 > gpurun_out/r05_gemm128_three_stage_ab3.log 2>&1; cat gpurun_out/r05_gemm128_three_stage_ab3.log
