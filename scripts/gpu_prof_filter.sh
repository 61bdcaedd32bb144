#!/bin/bash
# rocprofv3 kernel stats of the implicit-filter generation (fused vs PyTorch ops).  Usage: scripts/gpu_prof_filter.sh <tag> <L> [D]
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-flt}; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof -o flt -- python $R/scripts/bench_filter.py ${2:-1048576} ${3:-256} > $OUT/log.txt 2>&1
grep "^filter" $OUT/log.txt
python $R/scripts/rocpd_stats.py $(find $OUT/prof -name '*.db' | head -1) $OUT/flt_stats.csv | grep -E "hyena|Name" | cut -c1-170
find $OUT/prof -name '*.db' -size +30M -delete
