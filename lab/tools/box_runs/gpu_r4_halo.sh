#!/bin/bash
# first GPU call of the next round: lab/conv_halo_lab (LDS-halo direct conv prototype for the N = 128 layers) against a naive
# conv and next to the library's pf_conv3d_bf16.  Build locally first:
#   cd lab && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../include -I../pyramid-flow_amd/csrc conv_halo_lab.hip -ldl -o conv_halo_lab
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 120 lab/conv_halo_lab 2 64 64 1 > gpurun_out/r4_conv_halo_small.log 2>&1; cat gpurun_out/r4_conv_halo_small.log
timeout 300 lab/conv_halo_lab 8 256 256 1 > gpurun_out/r4_conv_halo.log 2>&1; cat gpurun_out/r4_conv_halo.log
