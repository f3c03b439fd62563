#!/bin/bash
# round 3, final tree: the whole GPU suite + smoke
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/r3_pytest_gpu_final.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r3_pytest_gpu_final.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3_smoke.log 2>&1; tail -2 gpurun_out/r3_smoke.log
