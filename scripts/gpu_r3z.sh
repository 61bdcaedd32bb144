#!/bin/bash
# whole GPU suite + smoke, then the model step / operator layer at the HyenaDNA lengths (bench.py lines without the CPU leg)
TAG=${1:-r3z}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
bash scripts/gpu_tests.sh $TAG
: > $OUT/sweep.jsonl
for cfg in "1048576 1 256" "32768 8 256" "160000 2 256" "450560 1 256" "1024 8 128"; do
    set -- $cfg
    timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --seq-len $1 --batch $2 --d-model $3 2>/dev/null | grep '^{"metric"' >> $OUT/sweep.jsonl
done
python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
print("| L | B | d | conv ms | HBM frac | layer ms | model ms (graphed) |")
for l in open(out + "/sweep.jsonl"):
    a = json.loads(l); c = a["config"]; m = a.get("model_step") or {}; g = (m.get("graphed") or {})
    print(f"| {c['seq_len']} | {c['batch_per_gpu']} | {c['channels']} | {a['ms_per_step']:.4f} | {a['roofline']['frac']:.3f} | "
          f"{a.get('operator_layer', {}).get('ms_per_step', 0):.3f} | {m.get('ms_per_step', 0):.2f} ({g.get('ms_per_step', 0) or 0:.2f}) |")
PY
