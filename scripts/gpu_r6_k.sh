#!/bin/bash
TAG=${1:-r6k}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for v in ip2prof ip2profnomem; do
  echo "== $v" | tee -a $OUT/phases.txt
  HYENA_FFTCONV_LIB=$R/build/libhyena_$v.so timeout 300 python scripts/ip2_phase_profile.py 1048576 1 256 2>&1 | tail -5 | tee -a $OUT/phases.txt
done
