#!/bin/bash
# HBM traffic counters (FETCH_SIZE, WRITE_SIZE: separate passes) of the projection / MLP kernels: scripts/gpu_pmc_proj.sh <tag> "<L B D>"
TAG=$1; CFG=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $OUT/$c -o pmc -- python $R/scripts/bench_proj.py "$CFG" > $OUT/$c.log 2>&1
  python $R/scripts/rocpd_pmc.py $(find $OUT/$c -name '*.db' | head -1) > $OUT/$c.csv 2>&1
  grep "hyena" $OUT/$c.csv | cut -c1-170
  find $OUT/$c -name '*.db' -delete
done
