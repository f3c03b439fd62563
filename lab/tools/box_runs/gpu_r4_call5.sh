#!/bin/bash
# round 4, call 5: the failures of call 4 with full traces, tr-read probe, wide halo conv tests + C5 A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 60 lab/tr_probe > gpurun_out/r4_tr_probe.log 2>&1; head -70 gpurun_out/r4_tr_probe.log
( timeout 900 python -m pytest tests/test_attention_w64_gpu.py tests/test_blocks_gpu.py "tests/test_sp_gpu.py::test_sp_engine_single_rank_matches_plain_engine" "tests/test_sp_gpu.py::test_sp_multi_process_exchange" tests/test_convhalo_gpu.py tests/test_vae_gpu.py -m gpu -q --durations=5 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo\|^E *$\|^$" | tail -150 ) > gpurun_out/r4_call5_pytest.log
cat gpurun_out/r4_call5_pytest.log | cut -c1-250
for pol in 6 -6; do
  ( timeout 300 python bench.py --workload c5_vae_768p_241f --steps 2 --warmup 1 --gemm-policy $pol 2>&1 | tail -1 ) > gpurun_out/r4_bench_c5_wide_$pol.log
  python - <<PY
import json
l=open("gpurun_out/r4_bench_c5_wide_$pol.log").read().strip().splitlines()[-1]
try:
    r=json.loads(l); print("C5 policy $pol:", r["value"], "frames/s", r["ms_per_step"], "ms", {k[11:]:(v["achieved"],v["ms_timed"],v["launches_timed"]) for k,v in r["roofline_other_kernels"].items() if "conv3d:" in k})
except Exception as e: print("C5 policy $pol: no JSON", l[-300:])
PY
done
