#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6p; mkdir -p $OUT
cd $R
timeout 600 python - <<PY 2>&1 | grep -v amdgpu | tee $OUT/real_shapes.txt
import json, torch, bench
dev = torch.device("cuda", 0)
for leg in bench.real_shape_legs(torch.bfloat16, dev):
    print(leg["seq_len"], leg["batch_per_gpu"], leg["d_model"], "operator", json.dumps(leg["operator_layer"])[:330])
    print("      model", json.dumps(leg["model_step"])[:420])
PY
