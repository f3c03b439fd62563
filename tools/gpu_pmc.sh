#!/bin/bash
# PMC passes over one kernel-only script. usage: gpu_pmc.sh <script.py> <tag>
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
REPO=$(pwd)
SCRIPT=$1; TAG=$2; ONLY=${3:-all}
cd /tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVES" \
            "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_VMEM" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  if [ "$ONLY" = "traffic" ] && [ $i -lt 4 ]; then continue; fi
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/pmc_${TAG}_$i -o p --output-format csv -- python $REPO/$SCRIPT 2 > /tmp/pmc_${TAG}_$i.log 2>&1
  f=$(find /tmp/pmc_${TAG}_$i -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python $REPO/tools/pmc_summarize.py $f >> $REPO/gpurun_out/pmc/${TAG}.txt; else echo "pass $i failed: $(tail -3 /tmp/pmc_${TAG}_$i.log)" >> $REPO/gpurun_out/pmc/${TAG}.txt; fi
done
cat $REPO/gpurun_out/pmc/${TAG}.txt
