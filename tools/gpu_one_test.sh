#!/bin/bash
# usage: gpu_one_test.sh <pytest args...>
mkdir -p gpurun_out
( timeout 1500 python -m pytest "$@" -x -q 2>&1 | tail -15 ) > gpurun_out/one_test.log
cat gpurun_out/one_test.log
