#!/bin/bash
# Shader-side PMC counters of the MFMA projection / MLP kernels (what are their wavefronts doing?): usage scripts/gpu_pmc_sq_proj.sh <tag> "<L B D>"
TAG=${1:-pmcsqpj}; CFG=${2:-"1048576 1 256"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES" "SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o pmc -- python $R/scripts/bench_proj.py "$CFG" > $OUT/p$i.log 2>&1
  echo "set $i rc=$?"
  python $R/scripts/rocpd_pmc.py $(find $OUT/p$i -name '*.db' | head -1) > $OUT/p$i.csv 2>&1
  find $OUT/p$i -name '*.db' -delete
done
cat $OUT/p*.csv | grep -E "mlp_kernel|inproj_pre" | cut -c1-160 | sort | tee $OUT/summary.csv
