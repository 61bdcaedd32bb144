#!/bin/bash
# A/B of library variants on one HyenaOperator layer: scripts/gpu_ab_op.sh <tag> "L B" ...
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for cfg in "$@"; do
  echo "== regular $cfg" | tee -a $OUT/ab_op.txt
  timeout 300 python scripts/bench_operator.py $cfg fused 2>&1 | tail -2 | tee -a $OUT/ab_op.txt
  for v in build/libhyena_*.so; do
    echo "== $v $cfg" | tee -a $OUT/ab_op.txt
    HYENA_FFTCONV_LIB=$R/$v timeout 300 python scripts/bench_operator.py $cfg fused 2>&1 | tail -2 | tee -a $OUT/ab_op.txt
  done
done
