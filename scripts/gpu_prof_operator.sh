#!/bin/bash
# rocprofv3 kernel stats of one HyenaOperator layer (fused mixer core vs PyTorch-glue path), diagnostic
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-opprof}; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof -o op -- python $R/scripts/bench_operator.py ${2:-1048576} ${3:-1} ${4:-fused} ${5:-256} > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt
python $R/scripts/rocpd_stats.py $(find $OUT/prof -name '*.db' | head -1) $OUT/op_stats.csv | head -45 | cut -c1-170
find $OUT/prof -name '*.db' -size +30M -delete
