#!/bin/bash
# round 5: the full -m gpu suite, then the bench line, then per-kernel profiles of one operator layer at 2^20 - 1 and 2^20
TAG=${1:-r5b}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("headline", d["ms_per_step"], d["roofline"]["frac"])
for r in d["sweep_real"]: print(r.get("seq_len"), r.get("ms_per_step"), r.get("aligned"), r.get("vs_aligned"), r.get("packed_ms"), r.get("error"))
for k in ("operator_layer","operator_layer_real","model_step","model_step_real"):
    r=d.get(k) or {}
    print(k, {x: r.get(x) for x in ("ms_per_step","min_ms","median_ms","vs_aligned","error")})
PY
if [ "$2" != "noprof" ]; then
for L in 1048575 1048576; do
  bash scripts/gpu_prof_operator.sh $TAG/op$L $L 1 fused 2>&1 | tail -3
done
fi
