#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of every kernel of the default bench command (C3).
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmcb_$ctr -o p --output-format csv -- python $REPO/bench.py --no-cpu-baseline > /tmp/pmcb_$ctr.log 2>&1
  f=$(find /tmp/pmcb_$ctr -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python $REPO/tools/pmc_summarize.py $f > $REPO/gpurun_out/pmc/bench_c3_$ctr.txt; else tail -5 /tmp/pmcb_$ctr.log > $REPO/gpurun_out/pmc/bench_c3_$ctr.txt; fi
  tail -2 /tmp/pmcb_$ctr.log | cut -c1-400 >> $REPO/gpurun_out/pmc/bench_c3_$ctr.txt
done
cat $REPO/gpurun_out/pmc/bench_c3_*.txt | cut -c1-200
