#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sp_gpu.py -m gpu -q -s -k "multi_process_exchange or launch_list" > gpurun_out/r3_sp_tests6.log 2>&1; echo "sp pytest rc=$?"; tail -3 gpurun_out/r3_sp_tests6.log
timeout 600 python bench.py > gpurun_out/r3_bench_c3_d.log 2> gpurun_out/r3_bench_c3_d.err; grep '^{' gpurun_out/r3_bench_c3_d.log | cut -c1-330
REPO=$(pwd); cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o c3 --output-format csv -- python $REPO/bench.py --no-cpu-baseline > $REPO/gpurun_out/r3_prof_c3_d.log 2>&1; echo "rocprof rc=$?"
find /tmp/prof_c3 -name '*kernel_stats*' -exec cp {} $REPO/gpurun_out/r3_c3_kernel_stats_d.csv \;
find /tmp/prof_c3 -name '*domain_stats*' -exec cp {} $REPO/gpurun_out/r3_c3_domain_stats_d.csv \;
head -12 $REPO/gpurun_out/r3_c3_kernel_stats_d.csv | cut -c1-160
grep '^{' $REPO/gpurun_out/r3_prof_c3_d.log | cut -c1-300
