#!/bin/bash
# round 3: PMC traffic passes (FETCH_SIZE / WRITE_SIZE / L2 hit) over one max-L forward and over one VAE tile-chunk window
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
rm -f gpurun_out/pmc/r3_forward_maxL.txt gpurun_out/pmc/r3_vae_tile.txt
bash tools/gpu_pmc.sh tools/forward_only.py r3_forward_maxL traffic > /dev/null 2>&1
bash tools/gpu_pmc.sh tools/vae_only.py r3_vae_tile traffic > /dev/null 2>&1
python tools/pmc_to_json.py gpurun_out/pmc/r3_forward_maxL.txt gpurun_out/r3_pmc_forward_maxL.json
python tools/pmc_to_json.py gpurun_out/pmc/r3_vae_tile.txt gpurun_out/r3_pmc_vae_tile.json
cp gpurun_out/pmc/r3_forward_maxL.txt gpurun_out/r3_pmc_forward_maxL.txt; cp gpurun_out/pmc/r3_vae_tile.txt gpurun_out/r3_pmc_vae_tile.txt
python - <<'PY'
import json
for f in ("gpurun_out/r3_pmc_forward_maxL.json", "gpurun_out/r3_pmc_vae_tile.json"):
    d = json.load(open(f))["kernels"]
    for k, v in d.items():
        print(f"{k[:64]:64s} n={v['launches']:4d} {v['hbm_bytes_per_launch']/1e6:9.1f} MB/launch  l2hit {v.get('l2_hit', float('nan')):.2f}")
PY
timeout 200 python tools/vae_bench.py 4:4 4:8 6:4 8:4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_vae_bench_lanes.log
