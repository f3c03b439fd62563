#!/bin/bash
# round 4, final artifacts of the committed tree: whole -m gpu suite + smoke, C3 bench line (+ cpu_baseline), rocprofv3 kernel
# stats of the same command, PMC traffic passes (max-L forward, VAE tile window, the dominant conv layer halo vs implicit GEMM)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/prof gpurun_out/pmc; export TMPDIR=/tmp
REPO=$(pwd)
( timeout 1500 python -m pytest tests -m gpu -q --durations=10 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo\|^$" | tail -45 ) > gpurun_out/r4_pytest_gpu_final.log
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 ) > gpurun_out/r4_smoke.log
tail -n 24 gpurun_out/r4_pytest_gpu_final.log | cut -c1-200; tail -n 1 gpurun_out/r4_smoke.log
( time timeout 900 python bench.py --steps 1 --warmup 1 ) > gpurun_out/r4_bench_c3_final.log 2>&1
tail -n 5 gpurun_out/r4_bench_c3_final.log | cut -c1-900
cd /tmp
( time timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o c3 --output-format csv -- python $REPO/bench.py --no-cpu-baseline ) > $REPO/gpurun_out/prof/r4_rocprof_c3.log 2>&1
find /tmp/prof_c3 -name '*stats*' -exec cp {} $REPO/gpurun_out/prof/ \;
cd $REPO
tail -3 gpurun_out/prof/r4_rocprof_c3.log | cut -c1-400
head -16 gpurun_out/prof/*kernel_stats.csv | cut -c1-170
rm -f gpurun_out/pmc/r4_forward_maxL.txt gpurun_out/pmc/r4_vae_tile.txt gpurun_out/pmc/r4_conv_layer.txt
bash tools/gpu_pmc.sh tools/forward_only.py r4_forward_maxL traffic > /dev/null 2>&1
bash tools/gpu_pmc.sh tools/vae_only.py r4_vae_tile traffic > /dev/null 2>&1
bash tools/gpu_pmc.sh tools/conv_only.py r4_conv_layer traffic > /dev/null 2>&1
python tools/pmc_to_json.py gpurun_out/pmc/r4_forward_maxL.txt gpurun_out/r04_pmc_forward_maxL.json
python tools/pmc_to_json.py gpurun_out/pmc/r4_vae_tile.txt gpurun_out/r04_pmc_vae_tile.json
python tools/pmc_to_json.py gpurun_out/pmc/r4_conv_layer.txt gpurun_out/r04_pmc_conv_layer_halo_vs_implicit_gemm.json
for t in forward_maxL vae_tile conv_layer; do cp gpurun_out/pmc/r4_$t.txt gpurun_out/r04_pmc_$t.txt; done
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_pmc_conv_layer_halo_vs_implicit_gemm.json"))["kernels"]
for k, v in d.items():
    print(f"{k[:72]:72s} n={v['launches']:4d} fetch(x2) {2*v['fetch_kb']/1e3:8.1f} MB write {v['write_kb']/1e3:8.1f} MB  l2hit {v.get('l2_hit', float('nan')):.2f}")
PY
