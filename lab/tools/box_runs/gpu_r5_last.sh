#!/bin/bash
# round 5, last call: the simplified epilogue (measurement options removed) -- harness A/B, GEMM tests, the C3 bench line of the final binary
mkdir -p gpurun_out
export TMPDIR=/tmp
GEMM_AB_SHAPES=0,1,2,3,4,5 timeout 200 tools/gemm_epi_ab 5 1000 1001 > gpurun_out/r05_gemm_epi_ab_final_binary.log 2>&1
cat gpurun_out/r05_gemm_epi_ab_final_binary.log
timeout 900 python -m pytest -x -q -m gpu tests/test_gemm8p_gpu.py tests/test_qk_epilogue_gpu.py tests/test_gemm_grouped_gpu.py "tests/test_fullsize_gpu.py::test_full_size_gemm_vs_library" "tests/test_fullsize_gpu.py::test_full_size_forward_vs_oracle_one_block_of_each_kind" tests/test_cmdlist_gpu.py tests/test_convhalo_gpu.py > gpurun_out/r05_gemm_tests_final_binary.log 2>&1
tail -3 gpurun_out/r05_gemm_tests_final_binary.log
( time timeout 900 python bench.py --steps 3 --warmup 1 ) > gpurun_out/r05_bench_c3_final.log 2>&1
python - <<'PY'
import json
ls=[l for l in open('gpurun_out/r05_bench_c3_final.log') if l.startswith('{')]
if ls:
    r=json.loads(ls[-1]); print('C3', r['value'], r['ms_per_step'], r['phases'], r['roofline']['achieved'], r['roofline']['frac'], r['whole_step_mfma_frac'], r['roofline'].get('traffic'))
    for k,v in r['roofline_family'].items(): print('   ', k, v['achieved'], v['frac'], v['ms_timed'], v.get('launches_timed'))
    for k,v in r['roofline_other_kernels'].items():
        if 'gemm8p' in k or 'halo' in k: print('   ', k, v['achieved'], v['ms_timed'], v.get('traffic'))
else:
    print(open('gpurun_out/r05_bench_c3_final.log').read()[-2500:])
PY
