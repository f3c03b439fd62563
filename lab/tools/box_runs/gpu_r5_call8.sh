#!/bin/bash
# round 5, GPU call 8: grouped-GEMM tests + the engines' tests; same-box A/B of the C3 video with / without grouped text GEMMs
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest -x -q -m gpu -s tests/test_gemm_grouped_gpu.py tests/test_gemm8p_gpu.py tests/test_qk_epilogue_gpu.py tests/test_flux_forward_gpu.py tests/test_cmdlist_gpu.py tests/test_blocks_gpu.py tests/test_fulldepth_oracle_gpu.py "tests/test_fullsize_gpu.py::test_full_size_forward_vs_oracle_one_block_of_each_kind" "tests/test_fullsize_gpu.py::test_mmdit_c4_length_forward_vs_oracle" tests/test_sp_gpu.py > gpurun_out/r05_grouped_gemm_tests.log 2>&1
tail -4 gpurun_out/r05_grouped_gemm_tests.log
grep -h "grouped vs two-stream" gpurun_out/r05_grouped_gemm_tests.log
for arm in off on; do
  timeout 600 python bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --group-text $arm > gpurun_out/r05_bench_c3_group_text_$arm.log 2>&1
  python - $arm <<'PY'
import json, sys
arm = sys.argv[1]
ls=[l for l in open(f'gpurun_out/r05_bench_c3_group_text_{arm}.log') if l.startswith('{')]
if ls:
    r=json.loads(ls[-1]); print('C3 group-text', arm, r['value'], r['ms_per_step'], r['phases'])
    for k,v in r['roofline_family'].items(): print('   ', k, v['achieved'], v['ms_timed'], v.get('launches_timed'))
else:
    print(open(f'gpurun_out/r05_bench_c3_group_text_{arm}.log').read()[-2500:])
PY
done
