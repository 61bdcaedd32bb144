#!/bin/bash
# Shader-side PMC counters of the headline bench (what are the wavefronts doing?): usage scripts/gpu_pmc_sq.sh <tag>
TAG=${1:-pmcsq}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES" "SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-operator > $OUT/p$i.log 2>&1
  echo "set $i rc=$?"
  python $R/scripts/rocpd_pmc.py $(find $OUT/p$i -name '*.db' | head -1) > $OUT/p$i.csv 2>&1
  find $OUT/p$i -name '*.db' -size +30M -delete
done
cat $OUT/p*.csv | grep -E "row_bwd|row_prod2_kernel|col_fwd_kernel<1024, 1>|col_inv_kernel<1024, 1>" | cut -c1-140
