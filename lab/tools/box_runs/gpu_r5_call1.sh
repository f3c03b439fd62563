#!/bin/bash
# round 5, GPU call 1: epilogue A/B of gemm8p, the new parity tests, the vendor-GEMM traffic diagnostic
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$(pwd)
timeout 120 tools/gemm_epi_ab 5 1000 1001 1002 1003 > gpurun_out/r05_gemm_epi_ab.log 2>&1
tail -12 gpurun_out/r05_gemm_epi_ab.log
timeout 900 python -m pytest -x -q -m gpu -s tests/test_gemm8p_gpu.py tests/test_qk_epilogue_gpu.py \
   "tests/test_fullsize_gpu.py::test_mmdit_c4_length_forward_vs_oracle" \
   "tests/test_fullsize_gpu.py::test_full_size_attention_properties_and_sampled_rows" \
   "tests/test_convhalo_gpu.py::test_halo_conv_upsampler_output_maps_equal_the_implicit_gemm" \
   > gpurun_out/r05_parity_edges_tests.log 2>&1
tail -15 gpurun_out/r05_parity_edges_tests.log
cd /tmp
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/pmc_blas_$i -o p --output-format csv -- python $REPO/tools/blas_pmc.py 3 > /tmp/pmc_blas_$i.log 2>&1
  f=$(find /tmp/pmc_blas_$i -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python $REPO/tools/pmc_runs.py $f >> $REPO/gpurun_out/r05_pmc_vendor_gemm.txt; else echo "pass $i failed: $(tail -3 /tmp/pmc_blas_$i.log)" >> $REPO/gpurun_out/r05_pmc_vendor_gemm.txt; fi
  k=$(find /tmp/pmc_blas_$i -name '*kernel_trace.csv' | head -1)
  if [ $i -eq 1 ] && [ -n "$k" ]; then python $REPO/tools/pmc_runs.py $k > $REPO/gpurun_out/r05_pmc_vendor_gemm_durations.txt 2>&1; fi
done
cat $REPO/gpurun_out/r05_pmc_vendor_gemm.txt
cat $REPO/gpurun_out/r05_pmc_vendor_gemm_durations.txt
