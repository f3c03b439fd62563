#!/bin/bash
# round 4, call 6: row-major V attention (tests + C3 line), fixed tests of call 4 / 5
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_attention_w64_gpu.py tests/test_fullsize_gpu.py tests/test_qk_epilogue_gpu.py tests/test_flux_forward_gpu.py tests/test_cmdlist_gpu.py tests/test_blocks_gpu.py tests/test_pipeline_gpu.py tests/test_fulldepth_oracle_gpu.py "tests/test_sp_gpu.py::test_sp_engine_single_rank_matches_plain_engine" "tests/test_sp_gpu.py::test_sp_multi_process_exchange" "tests/test_sp_gpu.py::test_sp_engine_launch_list_and_dead_rows_bit_identical_to_eager" -m gpu -q --durations=5 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo\|^E *$\|^$" | tail -120 ) > gpurun_out/r4_call6_pytest.log
cat gpurun_out/r4_call6_pytest.log | cut -c1-250
( timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 ) > gpurun_out/r4_bench_c3_call6.log
python - <<'PY'
import json
r=json.loads(open("gpurun_out/r4_bench_c3_call6.log").read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"], r.get("whole_step_mfma_frac"), r.get("phases"))
print({k:(v["achieved"],v["ms_timed"],v["launches_timed"]) for k,v in r["roofline_family"].items()})
print({k:(v["achieved"],v["ms_timed"]) for k,v in r["roofline_other_kernels"].items()})
PY
