#!/bin/bash
mkdir -p gpurun_out
( timeout 600 python tools/microbench.py gemm 2>&1 | tail -12 ) > gpurun_out/gemm_diag.log
cat gpurun_out/gemm_diag.log
