#!/bin/bash
# round 5, GPU call 7: grouped GEMM (text + image stream in one persistent launch); more co-residency probes; one C3 video
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest -x -q -m gpu -s tests/test_gemm_grouped_gpu.py tests/test_gemm8p_gpu.py tests/test_qk_epilogue_gpu.py tests/test_flux_forward_gpu.py tests/test_cmdlist_gpu.py tests/test_blocks_gpu.py tests/test_fulldepth_oracle_gpu.py "tests/test_fullsize_gpu.py::test_full_size_forward_vs_oracle_one_block_of_each_kind" "tests/test_fullsize_gpu.py::test_mmdit_c4_length_forward_vs_oracle" > gpurun_out/r05_grouped_gemm_tests.log 2>&1
tail -6 gpurun_out/r05_grouped_gemm_tests.log
grep -h "grouped vs two-stream" gpurun_out/r05_grouped_gemm_tests.log
COMM_OVERLAP_R=0,64,128 timeout 300 python tools/comm_overlap_bench.py 5 > gpurun_out/r05_comm_overlap_bench_r64_r128.log 2>&1
grep "RCCL" gpurun_out/r05_comm_overlap_bench_r64_r128.log | cut -c1-330
COMM_OVERLAP_R=0,16 NCCL_MAX_NCHANNELS=4 timeout 300 python tools/comm_overlap_bench.py 5 > gpurun_out/r05_comm_overlap_bench_4channels.log 2>&1
grep "RCCL" gpurun_out/r05_comm_overlap_bench_4channels.log | cut -c1-330
timeout 600 python bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r05_bench_c3_grouped.log 2>&1
python - <<'PY'
import json
ls=[l for l in open('gpurun_out/r05_bench_c3_grouped.log') if l.startswith('{')]
if ls:
    r=json.loads(ls[-1]); print('C3', r['value'], r['ms_per_step'], r['phases'])
    for k,v in r['roofline_family'].items(): print(' ', k, v['achieved'], v['ms_timed'], v.get('launches_timed'))
    for k,v in r['roofline_other_kernels'].items():
        if 'gemm8p' in k: print(' ', k, v['achieved'], v['ms_timed'])
else:
    print(open('gpurun_out/r05_bench_c3_grouped.log').read()[-2500:])
PY
