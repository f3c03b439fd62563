#!/bin/bash
mkdir -p gpurun_out
( timeout 600 python tools/dvfs_probe.py 2>&1 | tail -5 ) > gpurun_out/dvfs_probe.log
cat gpurun_out/dvfs_probe.log
