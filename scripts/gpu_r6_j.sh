#!/bin/bash
# round 6, call J: the full -m gpu suite, smoke(), the default bench line, model kernel stats (with / without the cast cache)
TAG=${1:-r6j}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $OUT/smoke.txt
timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("headline", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"])
for r in d["sweep"]: print(r.get("seq_len"), r.get("batch_per_gpu"), r.get("ms_per_step"), r.get("frac"), r.get("traffic"))
for k in ("operator_layer","operator_layer_real","model_step","model_step_real"):
    r=d.get(k) or {}
    print(k, {x: r.get(x) for x in ("ms_per_step","min_ms","median_ms","vs_aligned","error")})
for leg in d.get("real_shapes") or []:
    print(leg["seq_len"], leg["batch_per_gpu"], "operator", leg.get("operator_layer", {}).get("vs_aligned"), leg.get("operator_layer", {}).get("real", {}).get("min_ms"),
          "model", leg.get("model_step", {}).get("vs_aligned"), leg.get("model_step", {}).get("real", {}).get("min_ms"))
print("cpu", {k: d["cpu_baseline"].get(k) for k in ("value","cores","host_cores")})
PY
for cc in 1 0; do
  HYENA_CAST_CACHE=$cc timeout 300 python scripts/bench_model.py 1048576 1 256 8 2>&1 | tail -1 | cut -c1-220 | tee -a $OUT/model_castcache_ab.txt
done
