#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r3j; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_small.py tests/test_gpu_proj.py -q -m gpu > $OUT/pytest_new.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_new.txt
tail -5 $OUT/pytest_new.txt | cut -c1-220
echo "== bench_proj"; timeout 300 python scripts/bench_proj.py "1048576 1 256" 2>&1 | grep -v -i "warn\|amdgpu.ids" | tee $OUT/bench_proj.txt
for cfg in "1024 8 128" "1024 8 256" "2048 8 256"; do
  set -- $cfg
  timeout 400 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-operator --no-model --seq-len $1 --batch $2 --d-model $3 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('L=$1 B=$2 D=$3: conv ms', round(r['ms_per_step'], 5), 'frac', round(r['roofline']['frac'], 4), 'graph', r['config'].get('hipgraph_replay'))"
done
bash scripts/gpu_prof_bench.sh r3j_prof1k --seq-len 1024 --batch 8 --d-model 128 --no-operator --no-model --no-graph
