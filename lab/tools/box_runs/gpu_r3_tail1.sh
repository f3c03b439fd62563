#!/bin/bash
# round 3: gemm8p tail split -- parity tests, the A/B over every sequence length of a video; then (only if those pass) the
# whole GPU suite and the rocprofv3 kernel statistics of the bench command
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm8p_gpu.py -x -q -s > gpurun_out/r3_tail_tests.log 2>&1; rc=$?; echo "gemm8p pytest rc=$rc"; tail -5 gpurun_out/r3_tail_tests.log
[ $rc -ne 0 ] && exit 0
timeout 600 python tools/gemm_tail_ab.py 3 > gpurun_out/r3_gemm_tail_ab.log 2>&1; tail -12 gpurun_out/r3_gemm_tail_ab.log
timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/r3_full_tests2.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r3_full_tests2.log
REPO=$(pwd); cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o c3 --output-format csv -- python $REPO/bench.py --no-cpu-baseline > $REPO/gpurun_out/r3_prof_c3.log 2>&1; echo "rocprof rc=$?"
find /tmp/prof_c3 -name '*kernel_stats*' -exec cp {} $REPO/gpurun_out/r3_c3_kernel_stats.csv \;
find /tmp/prof_c3 -name '*domain_stats*' -exec cp {} $REPO/gpurun_out/r3_c3_domain_stats.csv \;
head -20 $REPO/gpurun_out/r3_c3_kernel_stats.csv | cut -c1-200
grep '^{' $REPO/gpurun_out/r3_prof_c3.log | cut -c1-300
