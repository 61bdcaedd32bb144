#!/bin/bash
# A/B timing of library variants on the projection / MLP micro-benchmarks: variants = build/libhyena_*.so + the regular library
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
echo "== regular" | tee $OUT/ab.txt
timeout 300 python scripts/bench_proj.py "$@" 2>&1 | grep "L=\|MLP" | tee -a $OUT/ab.txt
for v in build/libhyena_*.so; do
  echo "== $v" | tee -a $OUT/ab.txt
  HYENA_FFTCONV_LIB=$R/$v timeout 300 python scripts/bench_proj.py "$@" 2>&1 | grep "L=\|MLP" | tee -a $OUT/ab.txt
done
