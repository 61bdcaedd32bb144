#!/bin/bash
# experiment: occupancy vs spills in col_inv at M1 = 640 / 768 / 32 (HY_COLW); and the row0_bwd kernel without its spills
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4colw; mkdir -p $OUT
cd $R
for v in regular colw3 colw2; do
  for cfg in "655360 1" "786432 1" "40000 4"; do
    set -- $cfg
    if [ $v = regular ]; then L=""; else L="HYENA_FFTCONV_LIB=$R/build/libhyena_$v.so"; fi
    echo "== $v L=$1 B=$2" | tee -a $OUT/colw.txt
    env $L timeout 200 python bench.py --seq-len $1 --batch $2 --no-operator --no-model --no-sweep --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" | tee -a $OUT/colw.txt
  done
done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -2 | tee -a $OUT/colw.txt
