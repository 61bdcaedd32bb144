#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r5w; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_contract.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/pytest.txt
bash scripts/gpu_pmc_cfg.sh r5p_160k 160000 2 256 > /dev/null
bash scripts/gpu_pmc_cfg.sh r5p_450k 450560 1 256 > /dev/null
bash scripts/gpu_pmc_cfg.sh r5p_1m 1048576 1 256 > /dev/null
python - <<PY
import sys, json, torch
sys.path.insert(0, "$R")
import bench
dev = torch.device("cuda", 0)
r = bench.sweep_real_shapes(torch.bfloat16, "bf16", dev, 20, 3, True)
for x in r: print(x.get("seq_len"), round(x["ms_per_step"], 4), round(x["aligned"]["ms_per_step"], 4), round(x["vs_aligned"], 4), round(x["packed_ms"], 4))
json.dump(r, open("$OUT/sweep_real.json", "w"))
PY
