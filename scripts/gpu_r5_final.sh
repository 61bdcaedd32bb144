#!/bin/bash
# round 5, final: the full -m gpu suite, smoke(), the default bench line, then the operator at the B > 1 real shapes
TAG=${1:-r5final}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $OUT/smoke.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("headline", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["floor_frac"])
for r in d["sweep"]: print(r.get("seq_len"), r.get("batch_per_gpu"), r.get("ms_per_step"), r.get("frac"), r.get("traffic"))
for r in d["sweep_real"]: print(r.get("seq_len"), r.get("ms_per_step"), r.get("aligned"), r.get("vs_aligned"), r.get("packed_ms"), r.get("error"))
for k in ("operator_layer","operator_layer_real","model_step","model_step_real"):
    r=d.get(k) or {}
    print(k, {x: r.get(x) for x in ("ms_per_step","min_ms","median_ms","vs_aligned","error")})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
for cfg in "32767 8" "32768 8" "159999 2" "160000 2"; do
  timeout 300 python scripts/bench_operator.py $cfg fused 2>&1 | tail -1 | tee -a $OUT/op.txt
done
