#!/bin/bash
# final artifacts: plain C3 bench (JSON line) + rocprofv3 kernel stats of the same command
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
REPO=$(pwd)
( time timeout 900 python bench.py ) > gpurun_out/bench_c3.log 2>&1
tail -n 5 gpurun_out/bench_c3.log | cut -c1-1200
cd /tmp
( time timeout 1200 rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o c3 --output-format csv -- python $REPO/bench.py --no-cpu-baseline ) > $REPO/gpurun_out/prof/rocprof_c3.log 2>&1
find /tmp/prof_c3 -name '*stats*' -exec cp {} $REPO/gpurun_out/prof/ \;
tail -4 $REPO/gpurun_out/prof/rocprof_c3.log | cut -c1-600
head -14 $REPO/gpurun_out/prof/*kernel_stats.csv | cut -c1-160
