#!/bin/bash
# the 1k bench line (hipGraph replay of forward + backward) for the regular library and every build/libhyena_*.so, three runs each
TAG=${1:-ab1k}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
run() { timeout 100 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-operator --no-model --seq-len 1024 --batch 8 --d-model 128 2>/dev/null | grep '^{"metric"' | python -c "import sys,json; print('%.2f us' % (1e3 * json.loads(sys.stdin.read())['ms_per_step']))"; }
for i in 1 2 3; do
  echo "regular: $(run)" | tee -a $OUT/ab.txt
  for v in build/libhyena_*.so; do echo "$v: $(HYENA_FFTCONV_LIB=$R/$v run)" | tee -a $OUT/ab.txt; done
done
