#!/bin/bash
# round 4, call H: what the barrier before exchange 1 in dk's back-to-back transforms costs (regular = with it, nopresync = without)
TAG=${1:-r4h}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
CFGS='"32768 8 256" "16384 8 256" "8192 8 256" "4096 16 256" "2048 64 128" "32768 2 256"'
for v in regular nopresync regular nopresync regular nopresync; do
  echo "== $v" | tee -a $OUT/ab.txt
  if [ $v = regular ]; then unset HYENA_FFTCONV_LIB; else export HYENA_FFTCONV_LIB=$R/build/libhyena_$v.so; fi
  eval timeout 300 python scripts/oc_times.py $CFGS 2>&1 | grep "L=" | tee -a $OUT/ab.txt
done
unset HYENA_FFTCONV_LIB
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small.py -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest_gpu.txt
