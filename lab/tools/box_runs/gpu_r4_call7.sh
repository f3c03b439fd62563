#!/bin/bash
# round 4, call 7: two-taps-per-step variant of the halo conv in the lab harness (correctness + time next to the one-tap form)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 120 lab/conv_halo_lab 2 64 64 1 > gpurun_out/r4_conv_halo_t2_small.log 2>&1; cat gpurun_out/r4_conv_halo_t2_small.log
timeout 200 lab/conv_halo_lab 8 256 256 1 > gpurun_out/r4_conv_halo_t2.log 2>&1; cat gpurun_out/r4_conv_halo_t2.log
timeout 200 lab/conv_halo_lab 33 256 256 0 > gpurun_out/r4_conv_halo_t2_33f.log 2>&1; cat gpurun_out/r4_conv_halo_t2_33f.log
