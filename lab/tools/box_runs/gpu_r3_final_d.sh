#!/bin/bash
# last call of round 3: smoke + a few quick tests on the committed tree
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 100 python -m pytest tests/test_pipeline_gpu.py tests/test_hip_ops.py -m gpu -x -q 2>&1 | tail -2
