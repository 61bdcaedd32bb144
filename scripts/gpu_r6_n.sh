#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6n; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_cm.py tests/test_gpu_contract.py -m gpu -q -x -k "cm or mixer_core or lm_vs" 2>&1 | tail -3 | tee $OUT/pytest_cm.txt
bash scripts/gpu_prof_model.sh r6n_model_1023 1023 256 128 10 2 | grep -E "cm_|dgrad|^ms|value" | cut -c1-150
tail -1 gpurun_out/r6n_model_1023/log.txt | cut -c1-200
