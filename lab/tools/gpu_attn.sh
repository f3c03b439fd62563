#!/bin/bash
mkdir -p gpurun_out
for pr in 0 1 2 0 1 2; do echo "PF_ATTN_VAR=$pr"; PF_ATTN_VAR=$pr timeout 300 python tools/microbench.py attn 2>&1 | grep "prescaled=True"; done > gpurun_out/attn_prio.log
cat gpurun_out/attn_prio.log
