#!/bin/bash
# round 3, 16-bit filter kernels: GPU tests of everything that runs under autocast, smoke, filter timings (regular build and the
# two-workgroups-per-CU build of the backward kernels), the default bench line
TAG=${1:-r3v}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_filter.py tests/test_gpu_contract.py tests/test_gpu_block.py tests/test_gpu_proj.py tests/test_gpu_cm.py \
    tests/test_gpu_seqlen.py -q -m gpu --durations=8 -p no:cacheprovider > $OUT/pytest.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest.txt; tail -40 $OUT/pytest.txt
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.txt 2>&1; echo "smoke exit $?" >> $OUT/smoke.txt; tail -5 $OUT/smoke.txt
for L in 1048576 32768; do
  echo "== regular" | tee -a $OUT/filter.txt
  timeout 200 python scripts/bench_filter.py $L 256 2>&1 | grep "filter L" | tee -a $OUT/filter.txt
  for v in build/libhyena_*.so; do
    echo "== $v" | tee -a $OUT/filter.txt
    HYENA_FFTCONV_LIB=$R/$v timeout 200 python scripts/bench_filter.py $L 256 2>&1 | grep "filter L" | tee -a $OUT/filter.txt
  done
done
timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench exit $?"; tail -c 1500 $OUT/bench_default.json
