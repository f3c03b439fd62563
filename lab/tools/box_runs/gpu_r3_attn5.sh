#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 200 lab/attn_lab u5s1 2 ) 2>&1 | grep -v "^abl\|^   \|stamped" > gpurun_out/r3_attn_lab5.log
cat gpurun_out/r3_attn_lab5.log
