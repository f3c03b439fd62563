#!/bin/bash
# the test files from tests/test_fullsize_gpu.py on (the earlier ones passed on the same DiT code: r3_full_tests3.log), VAE tests first
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_vae_gpu.py -m gpu -x -q -s > gpurun_out/r3_vae_tests_gn.log 2>&1; rc=$?; echo "vae pytest rc=$rc"; tail -4 gpurun_out/r3_vae_tests_gn.log
[ $rc -ne 0 ] && exit 0
cd tests; FILES=$(ls test_*.py | awk '$0 >= "test_fullsize_gpu.py" && $0 != "test_vae_gpu.py"' | tr '\n' ' '); cd ..
timeout 1500 python -m pytest $(for f in $FILES; do echo tests/$f; done) -m gpu -x -q -s > gpurun_out/r3_full_tests4.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r3_full_tests4.log
timeout 600 python bench.py --workload c5_vae_768p_241f --no-cpu-baseline > gpurun_out/r3_bench_c5_gn.log 2>&1; grep '^{' gpurun_out/r3_bench_c5_gn.log | cut -c1-300
