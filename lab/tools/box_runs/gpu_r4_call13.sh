#!/bin/bash
# round 4, call 13: attn128 v2 (asm MFMAs on fixed accumulator registers) next to the shipped fast pass
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 200 lab/attn_lab u30s2 3; timeout 100 lab/attn_lab u5s1 3 ) 2>&1 | grep "^seq\|shipped attn\|alone\|pf_attention_bf16 with\|attn128 pipe" > gpurun_out/r4_attn128v2_lab.log
cat gpurun_out/r4_attn128v2_lab.log | cut -c1-220
