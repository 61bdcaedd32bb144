#!/bin/bash
# HBM traffic counters of the headline bench (separate passes: FETCH_SIZE and WRITE_SIZE do not fit one pass).
TAG=${1:-pmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $OUT/$c -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-operator > $OUT/$c.log 2>&1
  echo "$c rc=$?"
  python $R/scripts/rocpd_pmc.py $(find $OUT/$c -name '*.db' | head -1) > $OUT/$c.csv 2>&1
  grep hyena $OUT/$c.csv | cut -c1-160
  find $OUT/$c -name '*.db' -size +30M -delete
done
