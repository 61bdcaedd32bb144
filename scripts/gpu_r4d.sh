#!/bin/bash
# round 4, call D: self-paired rows of the two-level plan on a side stream -- A/B (HYENA_FFTCONV_SIDE), parity + hipGraph re-capture tests
TAG=${1:-r4d}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for side in 1 0 1 0; do
  for cfg in "160000 2" "450560 1" "1048576 1" "65536 4"; do
    set -- $cfg
    HYENA_FFTCONV_SIDE=$side timeout 300 python bench.py --seq-len $1 --batch $2 --steps 30 --warmup 5 --no-cpu-baseline --no-operator --no-model --no-sweep 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('side=$side L=$1 B=$2: %.4f ms/step frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))" | tee -a $OUT/ab.txt
  done
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_seqlen.py tests/test_gpu_contract.py -m gpu -x -q 2>&1 | tail -5 | tee $OUT/pytest_gpu.txt
