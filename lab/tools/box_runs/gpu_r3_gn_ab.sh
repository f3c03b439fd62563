#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/vae_bench.py 4:4 4:4:nofuse 4:4 4:4:nofuse > gpurun_out/r3_vae_gn_ab.log 2>&1; cat gpurun_out/r3_vae_gn_ab.log | grep n_streams
