#!/bin/bash
TAG=${1:-r5i}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $OUT/pytest.txt
for cfg in "32767 8" "32768 8" "159999 2" "160000 2" "1048575 1" "1048576 1"; do
  timeout 300 python scripts/bench_operator.py $cfg fused 2>&1 | tail -1 | tee -a $OUT/op.txt
done
