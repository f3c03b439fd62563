#!/bin/bash
# round 5, the budget's last minutes: tests that run the 128 x 128 kernel's conv form / K split and were not in the three-stage call's subset
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 170 python -m pytest -q -m gpu -p no:cacheprovider tests/test_encoder_fullwidth_gpu.py "tests/test_sp_gpu.py::test_vae_context_parallel" tests/test_reference_caller_gpu.py "tests/test_fulldepth_oracle_gpu.py::test_miniflux_full_depth_forward_vs_oracle" > gpurun_out/r05_tail_tests.log 2>&1
tail -5 gpurun_out/r05_tail_tests.log
