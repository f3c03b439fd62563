#!/bin/bash
# whole GPU suite + smoke on the final round-2 tree
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -m gpu -q --durations=5 2>&1 | grep -v "Warning\|^  " | tail -16 ) > gpurun_out/r2_pytest_gpu_final2.log
cat gpurun_out/r2_pytest_gpu_final2.log | cut -c1-200
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 ) > gpurun_out/r2_smoke_final2.log
cat gpurun_out/r2_smoke_final2.log
