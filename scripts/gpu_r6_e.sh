#!/bin/bash
# round 6, call E: what bounds the projection kernels -- the regular library against builds without the global stores / loads / both
TAG=${1:-r6e}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for v in regular nost nold nomem; do
  echo "== $v" | tee -a $OUT/ab.txt
  if [ $v = regular ]; then unset HYENA_FFTCONV_LIB; else export HYENA_FFTCONV_LIB=$R/build/libhyena_$v.so; fi
  timeout 300 python scripts/bench_inproj.py "1048576 1 256" "32768 8 256" 2>&1 | grep "gen \|L=" | cut -c1-110 | tee -a $OUT/ab.txt
  timeout 300 python scripts/bench_outproj.py "1048576 1 256" 2>&1 | grep "gen \|L=\|one kernel" | cut -c1-150 | tee -a $OUT/ab.txt
done
