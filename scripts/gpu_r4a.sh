#!/bin/bash
# round 4, call A: 16-byte row I/O A/B (conv / dk separately), full -m gpu suite on the new default, the default bench line (with `sweep`)
TAG=${1:-r4a}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for m in 0 1 2 3 0 3; do
  echo "== HYENA_FFTCONV_WIDE=$m" | tee -a $OUT/ab.txt
  HYENA_FFTCONV_WIDE=$m timeout 300 python scripts/oc_times.py "32768 8 256" "16384 8 256" "8192 8 256" "4096 16 256" "32768 2 256" "2048 64 128" 2>&1 | grep "L=" | tee -a $OUT/ab.txt
done
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 3000 $OUT/bench_default.json; tail -5 $OUT/bench_default.err
