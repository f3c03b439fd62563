#!/bin/bash
# round 4, call 1: halo conv prototype on hardware, attention row-sum variants (dot2 / pk_add / MFMA), new parity tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 120 lab/conv_halo_lab 2 64 64 1 > gpurun_out/r4_conv_halo_small.log 2>&1; cat gpurun_out/r4_conv_halo_small.log
timeout 200 lab/conv_halo_lab 8 256 256 1 > gpurun_out/r4_conv_halo.log 2>&1; cat gpurun_out/r4_conv_halo.log
( timeout 200 lab/attn_lab u30s2 3; timeout 100 lab/attn_lab u5s1 3 ) 2>&1 | grep "^seq\|shipped attn\|FAST\|library\|stamped\|S[0-3]:\|tail:\|wait\|K / Q\|total" > gpurun_out/r4_attn_lab.log
cat gpurun_out/r4_attn_lab.log | cut -c1-200
( timeout 900 python -m pytest tests/test_attention_w64_gpu.py tests/test_encoder_fullwidth_gpu.py tests/test_fullsize_gpu.py tests/test_i2v_gpu.py tests/test_capi.py -m "gpu or not gpu" -q -s --durations=8 2>&1 | grep -v "^$" | tail -40 ) > gpurun_out/r4_call1_pytest.log
cat gpurun_out/r4_call1_pytest.log | cut -c1-220
