#!/bin/bash
# launch-list tests, then the C3 video in eager / graph / eager / graph launch mode on the same box
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_cmdlist_gpu.py -x -q 2>&1 | tail -8 ) > gpurun_out/r2_cmdlist_tests3.log
cat gpurun_out/r2_cmdlist_tests3.log
out=gpurun_out/r2_bench_c3_launch_modes.log
: > $out
for m in eager graph eager graph; do
  echo "== --launch-mode $m" >> $out
  ( timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --launch-mode $m 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({k:d[k] for k in ('value','ms_per_step')}), d['config'].get('launch_mode'))" ) >> $out
done
cat $out
