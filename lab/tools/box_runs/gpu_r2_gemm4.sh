#!/bin/bash
# same-box A/B of several builds of the library, alternating processes.  usage: gpu_r2_gemm4.sh "lib1 lib2 ..." [shapes]
mkdir -p gpurun_out
out=gpurun_out/r2_gemm_ab4.log
: > $out
LIBS=${1:-"libpyflow_hip_old.so libpyflow_hip.so"}
SH=${2:-1,2,4}
for rep in 1 2; do
  for lib in $LIBS; do
    echo "== $lib (rep $rep)" >> $out
    ( GEMM_AB_SHAPES=$SH PF_BENCH_LIB=$lib timeout 300 python tools/gemm_ab.py 3 2>&1 | grep "^M=" | sed -e 's/(min [0-9]* max [0-9]*)//g' -e 's/\[auto\][^[]*//' -e 's/\[lib\].*//' ) >> $out
  done
done
cat $out
