#!/bin/bash
# round 6, call A: access-pattern probe, the new oracle-level tests, the default bench line with the real-shape legs, the copy/cast finder
TAG=${1:-r6a}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 120 build/proj_pattern_probe 2>&1 | tee $OUT/proj_pattern_probe.txt
timeout 1500 python -m pytest tests/test_gpu_contract.py -m gpu -q -k "forced or device_evaluated" 2>&1 | tail -15 | tee $OUT/pytest_forced.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "real_training_lengths" 2>&1 | tail -6 | tee $OUT/pytest_real_lengths.txt
timeout 300 python scripts/find_copies.py operator 1048575 1 > $OUT/copies_operator.txt 2>&1; tail -70 $OUT/copies_operator.txt
timeout 300 python scripts/find_copies.py model 1048576 1 2 > $OUT/copies_model.txt 2>&1; tail -5 $OUT/copies_model.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -3 $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("headline", d["ms_per_step"], d["roofline"]["frac"])
for r in d["sweep"]: print(r.get("seq_len"), r.get("batch_per_gpu"), r.get("ms_per_step"), r.get("frac"))
for k in ("operator_layer","operator_layer_real","model_step","model_step_real"):
    r=d.get(k) or {}
    print(k, {x: r.get(x) for x in ("ms_per_step","min_ms","median_ms","vs_aligned","error")})
for leg in d.get("real_shapes") or []:
    print(json.dumps(leg)[:900])
print("cpu", {k: d["cpu_baseline"].get(k) for k in ("value","cores","host_cores","by_threads","concurrent")})
PY
