#!/bin/bash
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_sp_gpu.py tests/test_vae_gpu.py -x -q -s 2>&1 | tail -25 ) > gpurun_out/sp_tests.log
cat gpurun_out/sp_tests.log
