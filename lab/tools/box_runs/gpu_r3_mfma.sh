#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 120 lab/mfma_shapes > gpurun_out/r3_mfma_shapes.log 2>&1; cat gpurun_out/r3_mfma_shapes.log
