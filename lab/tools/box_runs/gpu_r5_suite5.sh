#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest -q -m gpu -x --durations=10 tests/test_sp_gpu.py tests/test_text_encoder_gpu.py tests/test_vae_gpu.py tests/test_video_io.py tests/test_gemm_grouped_gpu.py > gpurun_out/r05_pytest_gpu_rest4.log 2>&1
tail -6 gpurun_out/r05_pytest_gpu_rest4.log
