#!/bin/bash
# round 4, call 12: attn128 experiment (one wave per SIMD, four blocks per wave, compiler-allocated AGPRs) next to the shipped fast pass
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 200 lab/attn_lab u30s2 3; timeout 100 lab/attn_lab u5s1 3 ) 2>&1 | grep "^seq\|shipped attn\|alone\|pf_attention_bf16 with" > gpurun_out/r4_attn128_lab.log
cat gpurun_out/r4_attn128_lab.log | cut -c1-220
