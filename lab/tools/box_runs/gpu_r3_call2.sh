#!/bin/bash
# round 3, call 2: attention laboratory (ablations + phase stamps), then the rest of the new parity tests
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 lab/attn_lab u30s2 4; timeout 120 lab/attn_lab u30s0 3 ) > gpurun_out/r3_attn_lab1.log 2>&1
cat gpurun_out/r3_attn_lab1.log
timeout 1500 python -m pytest -x -q -s -m gpu \
  "tests/test_fulldepth_oracle_gpu.py::test_vae_untiled_768p_chunked_vs_unchunked" \
  "tests/test_fulldepth_oracle_gpu.py::test_vae_released_width_three_latents_chunked_and_unchunked_vs_oracle" \
  tests/test_pipeline_gpu.py tests/test_reference_caller_gpu.py \
  "tests/test_vae_gpu.py::test_narrow_conv_every_channel_count_vs_conv3d" tests/test_bench_selflaunch_gpu.py \
  > gpurun_out/r3_parity_tests2.log 2>&1
echo "exit $?" >> gpurun_out/r3_parity_tests2.log
grep -v "^$" gpurun_out/r3_parity_tests2.log | tail -40
