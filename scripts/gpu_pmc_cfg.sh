#!/bin/bash
# HBM traffic counters (FETCH_SIZE, WRITE_SIZE: separate passes) of bench.py at one configuration:
#   scripts/gpu_pmc_cfg.sh <tag> <seq-len> <batch> <d-model>
TAG=$1; L=$2; B=$3; D=$4
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $OUT/$c -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-operator --no-model --no-sweep --no-graph --seq-len $L --batch $B --d-model $D > $OUT/$c.log 2>&1
  python $R/scripts/rocpd_pmc.py $(find $OUT/$c -name '*.db' | head -1) > $OUT/$c.csv 2>&1
  grep hyena $OUT/$c.csv | cut -c1-170
  find $OUT/$c -name '*.db' -delete
done
