#!/bin/bash
# round 3 final measurements: five-length sweep (full bench lines: conv, operator layer, model step) beside the unfused hipFFT path,
# kernel stats of the headline bench, of the workspace-free plan at 32k x 8 and 1k x 8, and of the model step
TAG=${1:-r3z}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
: > $OUT/sweep.jsonl; : > $OUT/unfused.jsonl
for cfg in "1024 8 128" "32768 8 256" "160000 2 256" "450560 1 256" "1048576 1 256"; do
    set -- $cfg
    timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --seq-len $1 --batch $2 --d-model $3 2>/dev/null | grep '^{"metric"' >> $OUT/sweep.jsonl
    timeout 300 python scripts/bench_unfused_gpu.py $1 $2 $3 bf16 2>/dev/null | grep '^{' >> $OUT/unfused.jsonl
done
python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
f = [json.loads(l) for l in open(out + "/sweep.jsonl")]
u = [json.loads(l) for l in open(out + "/unfused.jsonl")]
print("| L | B | d | conv ms | M nt/s | HBM frac | VALU frac | layer ms | model ms (graphed) | unfused hipFFT ms |")
for a, b in zip(f, u):
    c = a["config"]; m = a.get("model_step") or {}; g = (m.get("graphed") or {})
    print(f"| {c['seq_len']} | {c['batch_per_gpu']} | {c['channels']} | {a['ms_per_step']:.4f} | {a['value']/1e6:.1f} | {a['roofline']['frac']:.3f} | "
          f"{a['roofline_valu']['frac']:.3f} | {a.get('operator_layer', {}).get('ms_per_step', 0):.3f} | {m.get('ms_per_step', 0):.2f} ({g.get('ms_per_step', 0) or 0:.2f}) | {b['ms_per_step']:.2f} |")
PY
timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 600 $OUT/bench_default.json
bash scripts/gpu_prof_bench.sh ${TAG}_prof1m --no-operator --no-model
bash scripts/gpu_prof_bench.sh ${TAG}_prof32k --seq-len 32768 --batch 8 --no-operator --no-model
bash scripts/gpu_prof_bench.sh ${TAG}_prof1k --seq-len 1024 --batch 8 --d-model 128 --no-operator --no-model --no-graph
bash scripts/gpu_prof_model.sh ${TAG}_model 1048576 1 256 | head -30 | cut -c1-150
