#!/bin/bash
# A/B timing of the 16-bit filter kernels: scripts/gpu_ab_filter.sh <tag> -- variants = build/libhyena_*.so + the regular library
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for L in 1048576 160000 32768; do
  echo "== regular" | tee -a $OUT/ab.txt
  timeout 200 python scripts/bench_filter.py $L 256 --fused16-only 2>&1 | grep "filter L" | tee -a $OUT/ab.txt
  for v in build/libhyena_*.so; do
    echo "== $v" | tee -a $OUT/ab.txt
    HYENA_FFTCONV_LIB=$R/$v timeout 200 python scripts/bench_filter.py $L 256 --fused16-only 2>&1 | grep "filter L" | tee -a $OUT/ab.txt
  done
done
