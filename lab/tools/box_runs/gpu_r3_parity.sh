#!/bin/bash
# round 3, call 1: the new parity tests (full depth, bf16 trajectory, un-tiled VAE, from_pretrained/.pth vs oracle, narrow conv, self-launch)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest -x -q -s -m gpu \
  tests/test_fulldepth_oracle_gpu.py tests/test_pipeline_gpu.py tests/test_reference_caller_gpu.py \
  "tests/test_vae_gpu.py::test_narrow_conv_every_channel_count_vs_conv3d" tests/test_bench_selflaunch_gpu.py \
  > gpurun_out/r3_parity_tests.log 2>&1
echo "exit $?" >> gpurun_out/r3_parity_tests.log
tail -40 gpurun_out/r3_parity_tests.log
