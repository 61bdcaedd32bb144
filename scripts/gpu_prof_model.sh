#!/bin/bash
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o m -- python $R/scripts/bench_model.py "$@" > $OUT/log.txt 2>&1
python $R/scripts/rocpd_stats.py $(find $OUT/prof -name '*.db' | head -1) $OUT/stats.csv | head -45 | cut -c1-170
find $OUT/prof -name '*.db' -size +20M -delete
tail -2 $OUT/log.txt | cut -c1-300
