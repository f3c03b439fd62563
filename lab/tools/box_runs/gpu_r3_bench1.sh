#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python bench.py --steps 1 --warmup 0 > gpurun_out/r3_bench_c3_a.log 2> gpurun_out/r3_bench_c3_a.err
tail -3 gpurun_out/r3_bench_c3_a.err; python - <<'PY'
import json
l=[x for x in open('gpurun_out/r3_bench_c3_a.log') if x.startswith('{')][-1]
r=json.loads(l)
print(r['value'], r['ms_per_step'], r.get('peak_mem_gib'))
print(json.dumps(r['roofline']))
for k,v in r['roofline_other_kernels'].items(): print(k, v['achieved'], v['frac'], v['ms_timed'])
print(r['cpu_baseline']['value'])
PY
