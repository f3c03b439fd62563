#!/bin/bash
# round 4: lab/attn_lab with the dot2 row-sum variant of the FAST kernel (attention_w64.h MODE bit 3) next to the shipped pair
# (build lab/attn_lab locally first: see lab/README.md)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 200 lab/attn_lab u30s2 3; timeout 100 lab/attn_lab u5s1 3 ) 2>&1 | grep "^seq\|shipped attn\|FAST\|library" > gpurun_out/r4_attn_lab.log
cat gpurun_out/r4_attn_lab.log
