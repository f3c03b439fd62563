#!/bin/bash
# round 4, call 3: QK-RMSNorm + RoPE in the K|V|Q projection's epilogue: bitwise tests, forward / pipeline parity, C3 line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_qk_epilogue_gpu.py tests/test_hip_ops.py tests/test_flux_forward_gpu.py tests/test_cmdlist_gpu.py tests/test_blocks_gpu.py tests/test_fulldepth_oracle_gpu.py tests/test_pipeline_gpu.py tests/test_gemm8p_gpu.py -m gpu -q -x --durations=5 2>&1 | grep -v "^$" | tail -40 ) > gpurun_out/r4_call3_pytest.log
cat gpurun_out/r4_call3_pytest.log | cut -c1-220
( timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 ) > gpurun_out/r4_bench_c3_call3.log
python - <<'PY'
import json
r=json.loads(open("gpurun_out/r4_bench_c3_call3.log").read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"], r.get("whole_step_mfma_frac"))
print({k:(v["achieved"],v["ms_timed"],v["launches_timed"]) for k,v in r["roofline_family"].items()})
print({k:(v["achieved"],v["ms_timed"]) for k,v in r["roofline_other_kernels"].items()})
PY
