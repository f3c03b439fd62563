#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 200 lab/attn_lab u30s2 3; timeout 100 lab/attn_lab u30s0 3; timeout 100 lab/attn_lab u1s2 3; timeout 60 lab/attn_lab u5s1 3 ) 2>&1 | grep -v "^abl\|^   \|stamped" > gpurun_out/r3_attn_lab2.log
cat gpurun_out/r3_attn_lab2.log
