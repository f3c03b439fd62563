#!/bin/bash
# C3 video with the text stream on the side stream (default) vs on the compute stream, split-K on / off, same box
mkdir -p gpurun_out
out=gpurun_out/r2_bench_c3_text_stream_ab.log
: > $out
run() { echo "== $*" >> $out; ( timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({k:d[k] for k in ('value','ms_per_step')}))" ) >> $out; }
run
run --no-overlap-text
run
run --no-overlap-text
cat $out
