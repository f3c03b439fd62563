#!/bin/bash
# round 2 measurement set: C3 bench line (with cpu_baseline), PMC traffic of one max-L forward, rocprofv3 kernel stats of the bench command
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
REPO=$(pwd)
( timeout 900 python bench.py --steps 1 --warmup 0 2>gpurun_out/r2_bench_c3_c.err | tail -3 ) > gpurun_out/r2_bench_c3_c.log
cat gpurun_out/r2_bench_c3_c.log | cut -c1-3000; tail -3 gpurun_out/r2_bench_c3_c.err
rm -f gpurun_out/pmc/r2_forward_maxL.txt
bash tools/gpu_pmc.sh tools/forward_only.py r2_forward_maxL traffic > /dev/null 2>&1
python tools/pmc_to_json.py gpurun_out/pmc/r2_forward_maxL.txt gpurun_out/r2_pmc_forward_maxL.json
grep -A4 "gemm8p\|attn_kernel\|gemm256" gpurun_out/pmc/r2_forward_maxL.txt | head -60
cd /tmp
( time timeout 1500 rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o c3 --output-format csv -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline ) > $REPO/gpurun_out/r2_rocprof_c3.log 2>&1
find /tmp/prof_c3 -name '*kernel_stats*' -exec cp {} $REPO/gpurun_out/r2_c3_kernel_stats.csv \;
find /tmp/prof_c3 -name '*domain_stats*' -exec cp {} $REPO/gpurun_out/r2_c3_domain_stats.csv \;
head -24 $REPO/gpurun_out/r2_c3_kernel_stats.csv | cut -c1-220
tail -4 $REPO/gpurun_out/r2_rocprof_c3.log | cut -c1-600
