#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_i2v_gpu.py tests/test_text_encoder_gpu.py -x -q > gpurun_out/batch1_tests.log 2>&1
tail -n 15 gpurun_out/batch1_tests.log | cut -c1-300
timeout 300 python tools/blas_compare.py > gpurun_out/blas_compare.log 2>&1
tail -n 12 gpurun_out/blas_compare.log | cut -c1-300
