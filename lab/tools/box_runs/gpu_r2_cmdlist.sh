#!/bin/bash
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_cmdlist_gpu.py -x -q 2>&1 | tail -25 ) > gpurun_out/r2_cmdlist_tests.log
cat gpurun_out/r2_cmdlist_tests.log
( timeout 600 python tools/host_overhead.py 2>&1 | grep "forward" ) > gpurun_out/r2_host_overhead.log
cat gpurun_out/r2_host_overhead.log
