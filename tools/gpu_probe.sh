#!/bin/bash
mkdir -p gpurun_out
for gm in 4 2 8 16 1 4; do echo "GROUP_M=$gm"; PF_GEMM_GROUPM=$gm timeout 300 python tools/microbench.py gemm 2>&1 | grep -E "N=13440 K=1920|N=1920 K=7680|N=5760" | grep "M=30976" | sed 's/\[128x128\][^[]*//'; done > gpurun_out/groupm.log
cat gpurun_out/groupm.log
