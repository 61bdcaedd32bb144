#!/bin/bash
# A/B of library variants on scripts/bench_proj.py: bash scripts/gpu_r4_variants.sh <tag> <variant> [<variant> ...]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for v in "$@"; do
  echo "== $v" | tee -a $OUT/bench_proj.txt
  if [ "$v" = "regular" ]; then timeout 200 python scripts/bench_proj.py "1048576 1 256" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/bench_proj.txt
  else HYENA_FFTCONV_LIB=$R/build/libhyena_$v.so timeout 200 python scripts/bench_proj.py "1048576 1 256" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/bench_proj.txt; fi
done
