#!/bin/bash
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gemm256_gpu.py tests/test_hip_ops.py tests/test_flux_forward_gpu.py tests/test_vae_gpu.py -x -q 2>&1 | tail -3
