#!/bin/bash
# usage: gpu_one_test.sh <pytest args...>
mkdir -p gpurun_out
( timeout 1500 python -m pytest "$@" -q 2>&1 | grep -v "^  \|Warning" | tail -60 ) > gpurun_out/one_test.log
cat gpurun_out/one_test.log | cut -c1-400
