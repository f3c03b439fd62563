#!/bin/bash
# round 4, call 2: halo conv in the library (tests, C5 decode A/B), rank-shape prediction baseline, C3 bench line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_convhalo_gpu.py tests/test_vae_gpu.py tests/test_attention_w64_gpu.py tests/test_fullwidth_oracle_gpu.py -m gpu -q -s --durations=5 2>&1 | grep -v "^$" | tail -40 ) > gpurun_out/r4_call2_pytest.log
cat gpurun_out/r4_call2_pytest.log | cut -c1-220
for pol in 5 -5; do
  ( timeout 300 python bench.py --workload c5_vae_768p_241f --steps 2 --warmup 1 --gemm-policy $pol 2>&1 | tail -1 ) > gpurun_out/r4_bench_c5_policy_$pol.log
  python - <<PY
import json
l=open("gpurun_out/r4_bench_c5_policy_$pol.log").read().strip().splitlines()[-1]
try:
    r=json.loads(l); print("C5 policy $pol:", r["value"], "frames/s", r["ms_per_step"], "ms", {k:(v["achieved"],v["ms_timed"]) for k,v in r["roofline_other_kernels"].items()}, r["roofline"] and (r["roofline"]["achieved"]))
except Exception as e: print("C5 policy $pol: no JSON", l[-300:])
PY
done
( timeout 600 python tools/rank_shape_bench.py --out gpurun_out/r4_rank_shape_baseline.json 2>&1 | tail -140 ) > gpurun_out/r4_rank_shape_baseline.log
tail -24 gpurun_out/r4_rank_shape_baseline.log | cut -c1-250
( timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 ) > gpurun_out/r4_bench_c3_call2.log
cut -c1-1500 gpurun_out/r4_bench_c3_call2.log
