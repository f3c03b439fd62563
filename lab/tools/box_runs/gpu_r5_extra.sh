#!/bin/bash
# round 5: the other workloads of BASELINE.json on the final tree (C1 image, C2 384p, C4 image-to-video, C5 decode) and a sweep of the
# tail split's cost constant on the residual GEMM shapes
mkdir -p gpurun_out
export TMPDIR=/tmp
GEMM_AB_SHAPES=3,4,5,7 timeout 200 tools/gemm_epi_ab 5 0 402 403 406 408 > gpurun_out/r05_gemm_tail_overhead_sweep.log 2>&1
cat gpurun_out/r05_gemm_tail_overhead_sweep.log
for wl in c1_1024p_image c2_384p_121f c4_i2v_768p_121f c5_vae_768p_241f; do
  ( timeout 400 python bench.py --workload $wl --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 ) > gpurun_out/r05_bench_$wl.log
  python - <<PY
import json
l=open("gpurun_out/r05_bench_$wl.log").read().strip().splitlines()[-1]
try:
    r=json.loads(l); print("$wl:", r["value"], r["unit"], r["ms_per_step"], "ms", r.get("phases"), r["peak_mem_gib"], "GiB")
except Exception as e: print("$wl: no JSON", l[-400:])
PY
done
