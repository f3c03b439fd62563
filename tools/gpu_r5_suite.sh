#!/bin/bash
# the whole GPU tier, as the driver runs it at round end
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1700 python -m pytest tests/ -q -m gpu -x --durations=25 > gpurun_out/r05_pytest_gpu_full.log 2>&1
tail -45 gpurun_out/r05_pytest_gpu_full.log
