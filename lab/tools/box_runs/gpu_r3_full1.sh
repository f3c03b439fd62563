#!/bin/bash
# round 3: the whole GPU suite, the C3 bench line, and the rocprofv3 kernel statistics of the same bench command
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/r3_full_tests.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r3_full_tests.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3_smoke.log 2>&1; tail -2 gpurun_out/r3_smoke.log
timeout 600 python bench.py > gpurun_out/r3_bench_c3_b.log 2> gpurun_out/r3_bench_c3_b.err; tail -c 600 gpurun_out/r3_bench_c3_b.log
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r3_prof_c3" -o c3 -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/r3_prof_c3.log" 2>&1; echo "rocprof rc=$?"
cd "$GRAFT_REPO_ROOT"; find gpurun_out/r3_prof_c3 -name "*kernel_stats.csv" | head; find gpurun_out/r3_prof_c3 -name "*kernel_trace.csv" -delete; find gpurun_out/r3_prof_c3 -name "*.db" -delete
du -sh gpurun_out/r3_prof_c3
