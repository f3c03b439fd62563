#!/bin/bash
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_fullwidth_oracle_gpu.py::test_vae_default_width_tile_decode_vs_oracle tests/test_vae_gpu.py -x -q 2>&1 | tail -8 ) > gpurun_out/r2_narrow_tests.log
cat gpurun_out/r2_narrow_tests.log
( timeout 600 python bench.py --workload c5_vae_768p_241f --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({k:d[k] for k in ('value','ms_per_step')}))" ) > gpurun_out/r2_narrow_c5.log
cat gpurun_out/r2_narrow_c5.log
