#!/bin/bash
# round 6, final artefacts -> gpurun_out/r6final/ (copied into profiles/ afterwards): the C3 bench line with cpu_baseline and the
# in-run PMC traffic, the other configurations, rocprofv3 kernel stats of the bench command, PMC traffic passes over one max-L
# forward and one VAE tile-chunk window (the committed replay fall-back), cycle stamps of the final kernel
O=gpurun_out/r6final; mkdir -p $O gpurun_out/pmc
export TMPDIR=/tmp
REPO=$(pwd)
( time timeout 1500 python bench.py --steps 3 --warmup 1 ) > $O/r06_bench_c3_final.log 2>&1
python - <<'PY'
import json
ls=[l for l in open('gpurun_out/r6final/r06_bench_c3_final.log') if l.startswith('{')]
if ls:
    r=json.loads(ls[-1]); print('C3', r['value'], r['ms_per_step'], r['phases'], r['roofline']['achieved'], r['roofline']['frac'], r['whole_step_mfma_frac'], r['roofline'].get('traffic'), r['roofline'].get('traffic_note','')[:60], r.get('launch_lists'))
    for k,v in r['roofline_family'].items(): print('   ', k, v['achieved'], v['frac'], v['ms_timed'], v.get('launches_timed'))
    cb=r.get('cpu_baseline',{}); print('   cpu_baseline', cb.get('value'), cb.get('cores'), cb.get('fit_residuals'), cb.get('sample_spread'))
else:
    print(open('gpurun_out/r6final/r06_bench_c3_final.log').read()[-2500:])
PY
for w in c1_1024p_image c2_384p_121f c4_i2v_768p_121f c5_vae_768p_241f; do
  echo "# bench.py --workload $w --steps 2 --warmup 1" >> $O/r06_bench_c1_c2_c4_c5.log
  timeout 900 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | grep "^{" >> $O/r06_bench_c1_c2_c4_c5.log
done
python - <<'PY'
import json
for l in open('gpurun_out/r6final/r06_bench_c1_c2_c4_c5.log'):
    if l.startswith('{'):
        r=json.loads(l); print(r['config']['workload'][:28], r['value'], r['unit'], r['ms_per_step'], r.get('phases'))
PY
rm -f gpurun_out/pmc/r6_forward_maxL.txt gpurun_out/pmc/r6_vae_tile.txt
bash tools/gpu_pmc.sh tools/forward_only.py r6_forward_maxL traffic > /dev/null 2>&1
bash tools/gpu_pmc.sh tools/vae_only.py r6_vae_tile traffic > /dev/null 2>&1
python tools/pmc_to_json.py gpurun_out/pmc/r6_forward_maxL.txt $O/r06_pmc_forward_maxL.json
python tools/pmc_to_json.py gpurun_out/pmc/r6_vae_tile.txt $O/r06_pmc_vae_tile.json
cp gpurun_out/pmc/r6_forward_maxL.txt $O/r06_pmc_forward_maxL.txt; cp gpurun_out/pmc/r6_vae_tile.txt $O/r06_pmc_vae_tile.txt
python tools/gemm8p_stamps.py stamp1 > $O/r06_gemm8p_stamps_final_ktile.log 2>&1
python tools/gemm8p_stamps.py stamp3 > $O/r06_gemm8p_stamps_final_epilogue.log 2>&1
grep "steady\|boundary" $O/r06_gemm8p_stamps_final_ktile.log | cut -c1-200
cd /tmp
( time timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o c3 --output-format csv -- python $REPO/bench.py --no-cpu-baseline --no-pmc ) > $REPO/$O/r06_c3_rocprofv3_run.log 2>&1
f=$(find /tmp/prof_c3 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $REPO/$O/r06_c3_rocprofv3_kernel_stats.csv
f=$(find /tmp/prof_c3 -name '*domain_stats.csv' | head -1); [ -n "$f" ] && cp $f $REPO/$O/r06_c3_rocprofv3_domain_stats.csv
tail -2 $REPO/$O/r06_c3_rocprofv3_run.log | cut -c1-300
head -14 $REPO/$O/r06_c3_rocprofv3_kernel_stats.csv | cut -c1-150
