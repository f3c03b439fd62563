#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 120 lab/mfma_valu_overlap > gpurun_out/r3_mfma_valu_overlap.log 2>&1
cat gpurun_out/r3_mfma_valu_overlap.log
