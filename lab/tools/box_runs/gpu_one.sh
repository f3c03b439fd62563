#!/bin/bash
# usage: gpu_one.sh <pytest args...>
mkdir -p gpurun_out
timeout 900 python -m pytest "$@" -x -q > gpurun_out/one_full.log 2>&1
grep -n -m1 "Fatal Python error" gpurun_out/one_full.log && sed -n "$(grep -n -m1 'Fatal Python error' gpurun_out/one_full.log | cut -d: -f1),+25p" gpurun_out/one_full.log
tail -n 30 gpurun_out/one_full.log | cut -c1-300
