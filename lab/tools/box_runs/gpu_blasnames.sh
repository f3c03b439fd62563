#!/bin/bash
mkdir -p gpurun_out/blasprof
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/blasprof -o blas -- python /root/repo/tools/blas_names.py > /root/repo/gpurun_out/blasprof/run.log 2>&1
cd /root/repo
find gpurun_out/blasprof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -c1-600 {} | head -12'
find gpurun_out/blasprof -name "*kernel_trace.csv" -delete
