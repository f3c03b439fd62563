#!/bin/bash
# round 3, after the gemm8p tail split: the whole GPU suite, smoke, the C3 bench line and the rocprofv3 kernel statistics of the same command
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/r3_full_tests3.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r3_full_tests3.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3_smoke.log 2>&1; tail -2 gpurun_out/r3_smoke.log
timeout 600 python bench.py > gpurun_out/r3_bench_c3_c.log 2> gpurun_out/r3_bench_c3_c.err; grep '^{' gpurun_out/r3_bench_c3_c.log | cut -c1-400
REPO=$(pwd); cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o c3 --output-format csv -- python $REPO/bench.py --no-cpu-baseline > $REPO/gpurun_out/r3_prof_c3.log 2>&1; echo "rocprof rc=$?"
find /tmp/prof_c3 -name '*kernel_stats*' -exec cp {} $REPO/gpurun_out/r3_c3_kernel_stats.csv \;
find /tmp/prof_c3 -name '*domain_stats*' -exec cp {} $REPO/gpurun_out/r3_c3_domain_stats.csv \;
head -16 $REPO/gpurun_out/r3_c3_kernel_stats.csv | cut -c1-180
grep '^{' $REPO/gpurun_out/r3_prof_c3.log | cut -c1-300
