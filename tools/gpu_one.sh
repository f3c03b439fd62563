#!/bin/bash
# usage: gpu_one.sh <pytest args...>
mkdir -p gpurun_out
( timeout 900 python -m pytest "$@" -x -q 2>&1 | tail -30 ) > gpurun_out/one.log
cat gpurun_out/one.log
