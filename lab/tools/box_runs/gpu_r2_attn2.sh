#!/bin/bash
mkdir -p gpurun_out
( timeout 300 python tools/attn_bench.py 2>&1 | tail -12 ) > gpurun_out/r2_attn_bench2.log
cat gpurun_out/r2_attn_bench2.log
