#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_video_io.py tests/test_vae_gpu.py tests/test_text_encoder_gpu.py tests/test_gemm256_gpu.py tests/test_i2v_gpu.py tests/test_pipeline_gpu.py -x -q > gpurun_out/batch2_tests.log 2>&1
tail -n 8 gpurun_out/batch2_tests.log | cut -c1-300
( time timeout 900 python bench.py --steps 1 --warmup 0 ) > gpurun_out/bench_c3.log 2>&1
tail -n 5 gpurun_out/bench_c3.log | cut -c1-1800
