#!/bin/bash
# Shader-side PMC counters of the workspace-free kernels: usage scripts/gpu_pmc_oc.sh <tag> "<L B D>"
TAG=${1:-pmcoc}; CFG=${2:-"32768 8 256"}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES" "SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY" "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o pmc -- python $R/scripts/oc_times.py "$CFG" > $OUT/p$i.log 2>&1
  python $R/scripts/rocpd_pmc.py $(find $OUT/p$i -name '*.db' | head -1) > $OUT/p$i.csv 2>&1
  find $OUT/p$i -name '*.db' -delete
done
cat $OUT/p*.csv | grep -E "oc::" | cut -c1-150
