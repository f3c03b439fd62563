#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/gemm_slots.py > gpurun_out/gemm_slots.log 2>&1
tail -n 48 gpurun_out/gemm_slots.log | cut -c1-200
