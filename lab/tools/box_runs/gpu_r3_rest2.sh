#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_sp_gpu.py tests/test_text_encoder_gpu.py tests/test_video_io.py -m gpu -q -s > gpurun_out/r3_full_tests5.log 2>&1; echo "pytest rc=$?"
tail -6 gpurun_out/r3_full_tests5.log; dmesg 2>/dev/null | tail -5
