#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest -x -q -s -m gpu "tests/test_sp_gpu.py::test_vae_context_parallel" "tests/test_sp_gpu.py::test_vae_context_parallel_768p_two_ranks" tests/test_vae_gpu.py tests/test_fullwidth_oracle_gpu.py "tests/test_fulldepth_oracle_gpu.py::test_vae_untiled_768p_decode_vs_oracle" tests/test_i2v_gpu.py > gpurun_out/r3_cp_tests1.log 2>&1
echo "exit $?" >> gpurun_out/r3_cp_tests1.log
grep -v "^$" gpurun_out/r3_cp_tests1.log | tail -25
timeout 300 python tools/host_overhead.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3_host_overhead.log; cat gpurun_out/r3_host_overhead.log
