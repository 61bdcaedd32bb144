#!/bin/bash
# Throughput of the fused long convolution at the five HyenaDNA lengths (north_star: L in {1k,32k,160k,450k,1M}),
# beside the unfused torch.fft path on the same GPU.  Batch sizes per SURVEY.md 8d.  Usage: scripts/gpu_sweep.sh <tag>
tag=${1:-sweep}
out=gpurun_out/$tag
mkdir -p $out
: > $out/sweep.jsonl
: > $out/unfused.jsonl
for cfg in "1024 8 128" "32768 8 256" "160000 2 256" "450560 1 256" "1048576 1 256"; do
    set -- $cfg
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-operator --seq-len $1 --batch $2 --d-model $3 >> $out/sweep.jsonl
    timeout 300 python scripts/bench_unfused_gpu.py $1 $2 $3 bf16 >> $out/unfused.jsonl
done
python - "$out" <<'PY'
import json, sys
out = sys.argv[1]
f = [json.loads(l) for l in open(out + "/sweep.jsonl")]
u = [json.loads(l) for l in open(out + "/unfused.jsonl")]
print("| L | B | d | fused ms | fused M nt/s | roofline frac | unfused hipFFT ms | speed-up |")
print("|---|---|---|---|---|---|---|---|")
for a, b in zip(f, u):
    c = a["config"]
    print(f"| {c['seq_len']} | {c['batch_per_gpu']} | {c['channels']} | {a['ms_per_step']:.3f} | {a['value']/1e6:.1f} | "
          f"{a['roofline']['frac']:.3f} | {b['ms_per_step']:.2f} | {b['ms_per_step']/a['ms_per_step']:.1f}x |")
PY
