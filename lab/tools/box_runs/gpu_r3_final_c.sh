#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_vae_gpu.py tests/test_i2v_gpu.py tests/test_fullwidth_oracle_gpu.py -m gpu -x -q > gpurun_out/r3_conv_route_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3_conv_route_tests.log
