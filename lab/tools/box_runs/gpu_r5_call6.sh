#!/bin/bash
# round 5, GPU call 6: co-residency of communication kernels with the persistent GEMM; conv / gemm256 epilogue restructure;
# QK-norm + RoPE before the sequence-parallel exchange
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/comm_overlap_bench.py 7 > gpurun_out/r05_comm_overlap_bench.log 2>&1
cat gpurun_out/r05_comm_overlap_bench.log | tail -20
timeout 1200 python -m pytest -x -q -m gpu -s tests/test_gemm256_gpu.py tests/test_convhalo_gpu.py tests/test_vae_gpu.py tests/test_fullwidth_oracle_gpu.py tests/test_qk_epilogue_gpu.py tests/test_sp_gpu.py > gpurun_out/r05_conv_gemm256_sp_tests.log 2>&1
tail -4 gpurun_out/r05_conv_gemm256_sp_tests.log
timeout 600 python bench.py --workload c5_vae_768p_241f --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r05_bench_c5_epilogues.log 2>&1
python - <<'PY'
import json
ls=[l for l in open('gpurun_out/r05_bench_c5_epilogues.log') if l.startswith('{')]
if ls:
    r=json.loads(ls[-1]); print('C5', r['value'], r['ms_per_step'], r.get('roofline'))
else:
    print(open('gpurun_out/r05_bench_c5_epilogues.log').read()[-1500:])
PY
