#!/bin/bash
TAG=${1:-r6h}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for v in nomem nofrag nostag; do
  echo "== $v" | tee -a $OUT/ab.txt
  HYENA_FFTCONV_LIB=$R/build/libhyena_$v.so GEN=2 timeout 300 python scripts/bench_inproj.py "1048576 1 256" 2>&1 | grep "gen \|L=" | cut -c1-110 | tee -a $OUT/ab.txt
done
