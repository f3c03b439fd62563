#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command (C3) ; only the stats summaries come back.
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp
( time timeout 1500 rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o c3 --output-format csv -- python $REPO/bench.py --no-cpu-baseline ) > $REPO/gpurun_out/prof/rocprof_c3.log 2>&1
find /tmp/prof_c3 -name '*stats*' -exec cp {} $REPO/gpurun_out/prof/ \;
ls -la /tmp/prof_c3/* | head -20 >> $REPO/gpurun_out/prof/rocprof_c3.log
tail -5 $REPO/gpurun_out/prof/rocprof_c3.log
head -40 $REPO/gpurun_out/prof/*kernel_stats.csv
