#!/bin/bash
# round 4: PMC traffic passes (FETCH_SIZE / WRITE_SIZE / L2 hit; own runs, kernel trace only) over one max-L forward and one VAE
# tile-chunk window of the FINAL tree -> profiles/r04_pmc_*.json (what bench.py's `traffic` fields are replayed from)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
rm -f gpurun_out/pmc/r4_forward_maxL.txt gpurun_out/pmc/r4_vae_tile.txt
bash tools/gpu_pmc.sh tools/forward_only.py r4_forward_maxL traffic > /dev/null 2>&1
bash tools/gpu_pmc.sh tools/vae_only.py r4_vae_tile traffic > /dev/null 2>&1
python tools/pmc_to_json.py gpurun_out/pmc/r4_forward_maxL.txt gpurun_out/r04_pmc_forward_maxL.json
python tools/pmc_to_json.py gpurun_out/pmc/r4_vae_tile.txt gpurun_out/r04_pmc_vae_tile.json
cp gpurun_out/pmc/r4_forward_maxL.txt gpurun_out/r04_pmc_forward_maxL.txt; cp gpurun_out/pmc/r4_vae_tile.txt gpurun_out/r04_pmc_vae_tile.txt
python - <<'PY'
import json
for f in ("gpurun_out/r04_pmc_forward_maxL.json", "gpurun_out/r04_pmc_vae_tile.json"):
    d = json.load(open(f))["kernels"]
    for k, v in d.items():
        print(f"{k[:72]:72s} n={v['launches']:4d} {v['hbm_bytes_per_launch']/1e6:9.1f} MB/launch  l2hit {v.get('l2_hit', float('nan')):.2f}")
PY
