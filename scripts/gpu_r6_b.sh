#!/bin/bash
# round 6, call B: row-piece probe; the real-shape legs with the model-level sequence padding
TAG=${1:-r6b}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 200 build/rowpiece_probe 2>&1 | tee $OUT/rowpiece_probe.txt
timeout 600 python - <<PY 2>&1 | tee $OUT/real_shapes.txt
import json, torch, bench
dev = torch.device("cuda", 0)
legs = bench.real_shape_legs(torch.bfloat16, dev, no_operator=True)
for leg in legs:
    m = leg["model_step"]
    print(leg["seq_len"], leg["batch_per_gpu"], leg["d_model"], json.dumps(m)[:700])
PY
