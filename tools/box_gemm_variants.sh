#!/bin/bash
# same-box A/B of library builds (pyramid-flow_amd/variants/<name>/libpyflow_hip.so) on the DiT's GEMM shapes, alternating
# processes: tools/box_gemm_variants.sh <rounds> <variant> <variant> ...   ("ship" = the shipping library)
cd /root/repo
R=$1; shift
for rep in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = ship ]; then LP=pyramid-flow_amd; else LP=pyramid-flow_amd/variants/$v; fi
    echo "### rep $rep variant $v"
    GEMM_AB_SHAPES=${GEMM_AB_SHAPES:-0,1,2,3,4,5} LD_LIBRARY_PATH=$LP:$LD_LIBRARY_PATH tools/gemm_epi_ab $R 0
  done
done
