#!/bin/bash
# the rebuilt library loads and runs on the GPU box: smoke() + C-API + launch-list tests
mkdir -p gpurun_out
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1; timeout 300 python -m pytest tests/test_capi.py tests/test_cmdlist_gpu.py -q 2>&1 | tail -2 ) > gpurun_out/sanity.log
cat gpurun_out/sanity.log
