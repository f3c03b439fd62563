#!/bin/bash
# round 4, call 9: halo conv with the upsamplers' output maps + XCD-aware order: tests, C5 A/B, PMC of the VAE tile window
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_convhalo_gpu.py tests/test_vae_gpu.py tests/test_fullwidth_oracle_gpu.py::test_vae_default_width_tile_decode_vs_oracle tests/test_fulldepth_oracle_gpu.py::test_vae_untiled_768p_decode_vs_oracle -m gpu -q --durations=3 2>&1 | grep -v "amdgpu.ids\|^$" | tail -40 ) > gpurun_out/r4_call9_pytest.log
cat gpurun_out/r4_call9_pytest.log | cut -c1-230
for pol in 7 -7; do
  ( timeout 300 python bench.py --workload c5_vae_768p_241f --steps 2 --warmup 1 --gemm-policy $pol 2>&1 | tail -1 ) > gpurun_out/r4_bench_c5_maps_$pol.log
  python - <<PY
import json
l=open("gpurun_out/r4_bench_c5_maps_$pol.log").read().strip().splitlines()[-1]
try:
    r=json.loads(l); print("C5 policy $pol:", r["value"], "frames/s", r["ms_per_step"], "ms", {k[11:]:(v["achieved"],v["ms_timed"],v["launches_timed"]) for k,v in r["roofline_other_kernels"].items() if "conv3d:" in k})
except Exception as e: print("C5 policy $pol: no JSON", l[-300:])
PY
done
rm -f gpurun_out/pmc/r4_vae_tile.txt
bash tools/gpu_pmc.sh tools/vae_only.py r4_vae_tile traffic > /dev/null 2>&1
python tools/pmc_to_json.py gpurun_out/pmc/r4_vae_tile.txt gpurun_out/r04_pmc_vae_tile.json
cp gpurun_out/pmc/r4_vae_tile.txt gpurun_out/r04_pmc_vae_tile.txt
grep -A2 "conv_halo128\|gemm8p_kernel<true" gpurun_out/r04_pmc_vae_tile.txt | head -30
