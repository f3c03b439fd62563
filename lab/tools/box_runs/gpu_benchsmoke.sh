#!/bin/bash
mkdir -p gpurun_out
( timeout 300 python bench.py --workload smoke_128p_17f --tiny-model --no-cpu-baseline 2>&1 | tail -3 ) > gpurun_out/bench_1_tiny.log
python - <<'PY'
import json
l=[x for x in open('gpurun_out/bench_1_tiny.log') if x.startswith('{')]
print(open('gpurun_out/bench_1_tiny.log').read()[-600:] if not l else {k:(v['achieved'],v['unit'],v['launches_timed']) for k,v in json.loads(l[0])['roofline_other_kernels'].items()})
PY
