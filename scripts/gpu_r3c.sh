#!/bin/bash
# round 3, GPU call 3: re-test the small / projection kernels after the fixes, micro-benchmarks, bench lines
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-r3c}; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_small.py tests/test_gpu_proj.py tests/test_gpu_cm.py -q -m gpu -x > $OUT/pytest_new.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_new.txt
tail -12 $OUT/pytest_new.txt | cut -c1-220
echo "== bench_proj"; timeout 300 python scripts/bench_proj.py "1048576 1 256" "32768 8 256" 2>&1 | grep -v -i "warn\|amdgpu.ids" | tee $OUT/bench_proj.txt
for cfg in "1048576 1 256" "1024 8 128" "32768 8 256"; do
  set -- $cfg
  timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --seq-len $1 --batch $2 --d-model $3 > $OUT/bench_$1.json 2> $OUT/bench_$1.err
  python - <<PY
import json
r = json.loads([l for l in open("$OUT/bench_$1.json") if l.startswith("{")][-1])
m = r.get("model_step") or {}
print("L=$1: conv ms", round(r["ms_per_step"], 5), "frac", round(r["roofline"]["frac"], 4), "graph", r["config"].get("hipgraph_replay"), "| layer", r.get("operator_layer", {}).get("ms_per_step"), "| model", m.get("ms_per_step"), m.get("error"), (m.get("graphed") or {}).get("ms_per_step"), (m.get("graphed") or {}).get("error"))
PY
done
