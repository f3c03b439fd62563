#!/bin/bash
# round 5, GPU call 2: gemm8p with the compiler's hidden store drains removed, LDS ring 8 / 10, pre-issue; parity tests
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 tools/gemm_epi_ab 5 1108,1000 1108,1001 1108,1003 1108,1007 1110,1003 1110,1007 > gpurun_out/r05_gemm_epi_ab2.log 2>&1
cat gpurun_out/r05_gemm_epi_ab2.log
timeout 900 python -m pytest -x -q -m gpu -s tests/test_gemm8p_gpu.py tests/test_qk_epilogue_gpu.py \
   "tests/test_fullsize_gpu.py::test_mmdit_c4_length_forward_vs_oracle" \
   "tests/test_fullsize_gpu.py::test_full_size_attention_properties_and_sampled_rows" \
   "tests/test_fullsize_gpu.py::test_full_size_gemm_vs_library" \
   "tests/test_convhalo_gpu.py::test_halo_conv_upsampler_output_maps_equal_the_implicit_gemm" \
   > gpurun_out/r05_parity_edges_tests.log 2>&1
tail -5 gpurun_out/r05_parity_edges_tests.log
grep -h "rel-L2\|L = 1\|token-major\|upsampler" gpurun_out/r05_parity_edges_tests.log | head -20
