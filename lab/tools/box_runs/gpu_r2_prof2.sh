#!/bin/bash
# rocprofv3 kernel stats of the bench command on the final round-2 tree
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp
( time timeout 1500 rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o c3 --output-format csv -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline ) > $REPO/gpurun_out/r2_rocprof_c3_final.log 2>&1
find /tmp/prof_c3 -name '*kernel_stats*' -exec cp {} $REPO/gpurun_out/r2_c3_kernel_stats_final.csv \;
find /tmp/prof_c3 -name '*domain_stats*' -exec cp {} $REPO/gpurun_out/r2_c3_domain_stats_final.csv \;
head -24 $REPO/gpurun_out/r2_c3_kernel_stats_final.csv | cut -c1-200
tail -4 $REPO/gpurun_out/r2_rocprof_c3_final.log | cut -c1-500
