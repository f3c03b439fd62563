#!/bin/bash
# round 5, final artefacts: the C3 bench line (with cpu_baseline), rocprofv3 kernel stats of the bench command, PMC traffic passes
# (own runs, kernel trace only) over one max-L forward and one VAE tile-chunk window -> profiles/r05_*
mkdir -p gpurun_out/prof gpurun_out/pmc
export TMPDIR=/tmp
REPO=$(pwd)
( time timeout 900 python bench.py --steps 3 --warmup 1 ) > gpurun_out/r05_bench_c3_final.log 2>&1
python - <<'PY'
import json
ls=[l for l in open('gpurun_out/r05_bench_c3_final.log') if l.startswith('{')]
if ls:
    r=json.loads(ls[-1]); print('C3', r['value'], r['ms_per_step'], r['phases'], r['roofline']['achieved'], r['roofline']['frac'], r['whole_step_mfma_frac'])
    for k,v in r['roofline_family'].items(): print('   ', k, v['achieved'], v['frac'], v['ms_timed'], v.get('launches_timed'))
    for k,v in r['roofline_other_kernels'].items():
        if 'gemm8p' in k or 'halo' in k: print('   ', k, v['achieved'], v['ms_timed'])
    print('   cpu_baseline', r['cpu_baseline']['value'], r['cpu_baseline']['cores'])
else:
    print(open('gpurun_out/r05_bench_c3_final.log').read()[-2500:])
PY
rm -f gpurun_out/pmc/r5_forward_maxL.txt gpurun_out/pmc/r5_vae_tile.txt
bash tools/gpu_pmc.sh tools/forward_only.py r5_forward_maxL traffic > /dev/null 2>&1
bash tools/gpu_pmc.sh tools/vae_only.py r5_vae_tile traffic > /dev/null 2>&1
python tools/pmc_to_json.py gpurun_out/pmc/r5_forward_maxL.txt gpurun_out/r05_pmc_forward_maxL.json
python tools/pmc_to_json.py gpurun_out/pmc/r5_vae_tile.txt gpurun_out/r05_pmc_vae_tile.json
cp gpurun_out/pmc/r5_forward_maxL.txt gpurun_out/r05_pmc_forward_maxL.txt; cp gpurun_out/pmc/r5_vae_tile.txt gpurun_out/r05_pmc_vae_tile.txt
python - <<'PY'
import json
for f in ("gpurun_out/r05_pmc_forward_maxL.json", "gpurun_out/r05_pmc_vae_tile.json"):
    d = json.load(open(f))["kernels"]
    for k, v in d.items():
        print(f"{k[:72]:72s} n={v['launches']:4d} {v['hbm_bytes_per_launch']/1e6:9.1f} MB/launch  l2hit {v.get('l2_hit', float('nan')):.2f}")
PY
cd /tmp
( time timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o c3 --output-format csv -- python $REPO/bench.py --no-cpu-baseline ) > $REPO/gpurun_out/prof/r05_rocprof_c3.log 2>&1
find /tmp/prof_c3 -name '*stats*' -exec cp {} $REPO/gpurun_out/prof/ \;
tail -3 $REPO/gpurun_out/prof/r05_rocprof_c3.log | cut -c1-400
head -16 $REPO/gpurun_out/prof/*kernel_stats.csv | cut -c1-170
