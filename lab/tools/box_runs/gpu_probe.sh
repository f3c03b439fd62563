#!/bin/bash
mkdir -p gpurun_out
( timeout 600 python tools/host_overhead.py 2>&1 | tail -5 ) > gpurun_out/host_overhead.log
cat gpurun_out/host_overhead.log
