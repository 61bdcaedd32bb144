#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r2l; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_cm.py -q -m gpu -x > $OUT/pytest_cm.txt 2>&1; echo "exit $?" >> $OUT/pytest_cm.txt
for lay in channel position; do
  for cfg in "1048576 1" "32768 8" "160000 2"; do set -- $cfg
    HYENA_MIXER_LAYOUT=$lay timeout 300 python scripts/bench_operator.py $1 $2 fused 2>&1 | tail -1 | sed "s/^/$lay L=$1 B=$2: /" >> $OUT/operator.txt
  done
done
tail -4 $OUT/pytest_cm.txt; cat $OUT/operator.txt
