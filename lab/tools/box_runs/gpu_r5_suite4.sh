#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest -q -m gpu -x --durations=10 tests/test_sp_gpu.py tests/test_text_encoder_gpu.py tests/test_vae_gpu.py tests/test_video_io.py > gpurun_out/r05_pytest_gpu_rest3.log 2>&1
tail -6 gpurun_out/r05_pytest_gpu_rest3.log
timeout 900 python tools/rank_shape_bench.py --out gpurun_out/r05_rank_shape_prediction.json > gpurun_out/r05_rank_shape_prediction.log 2>&1
tail -16 gpurun_out/r05_rank_shape_prediction.log | cut -c1-260
