#!/bin/bash
# one-launch kernels for short rows with the independent transforms of a row on separate wavefronts: timings (and the general kernels
# for comparison), the 1k bench line, then the whole GPU suite + smoke
TAG=${1:-r3ad}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
echo "== one launch per direction (filter / u transforms on their own wavefronts)" | tee $OUT/small.txt
timeout 200 python scripts/oc_times.py "1024 8 128" "1024 8 256" "2048 4 256" "2048 8 256" "1024 16 128" 2>&1 | grep "L=" | tee -a $OUT/small.txt
echo "== HYENA_FFTCONV_SMALL=0: spec + conv, conv + dk" | tee -a $OUT/small.txt
HYENA_FFTCONV_SMALL=0 timeout 200 python scripts/oc_times.py "1024 8 128" "2048 4 256" 2>&1 | grep "L=" | tee -a $OUT/small.txt
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-operator --no-model --seq-len 1024 --batch 8 --d-model 128 2>/dev/null | grep '^{"metric"' > $OUT/bench_1k.json
python -c "import json,sys; a=json.load(open('$OUT/bench_1k.json')); print('1k bench line: ms_per_step', a['ms_per_step'], 'frac', a['roofline']['frac'])"
bash scripts/gpu_tests.sh $TAG
