#!/bin/bash
# round 4, call 10: the other workloads of BASELINE.json on the final tree (C1 image, C2 384p, C4 image-to-video, C5 decode) and the
# rank-shape prediction of the final tree
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for wl in c1_1024p_image c2_384p_121f c4_i2v_768p_121f c5_vae_768p_241f; do
  ( timeout 400 python bench.py --workload $wl --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 ) > gpurun_out/r4_bench_$wl.log
  python - <<PY
import json
l=open("gpurun_out/r4_bench_$wl.log").read().strip().splitlines()[-1]
try:
    r=json.loads(l); print("$wl:", r["value"], r["unit"], r["ms_per_step"], "ms", r.get("phases"), r["peak_mem_gib"], "GiB")
except Exception as e: print("$wl: no JSON", l[-400:])
PY
done
timeout 700 python tools/rank_shape_bench.py --out gpurun_out/r4_rank_shape_final.json > gpurun_out/r4_rank_shape_final.log 2>&1
tail -16 gpurun_out/r4_rank_shape_final.log | cut -c1-260
