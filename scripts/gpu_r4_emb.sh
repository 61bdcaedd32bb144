#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4emb; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_block.py tests/test_gpu_proj.py tests/test_gpu_runner.py tests/test_gpu_contract.py -q -m gpu 2>&1 | tail -3 | tee $OUT/pytest.txt
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2 | tee $OUT/smoke.txt
timeout 400 python bench.py --no-sweep --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('headline', d['ms_per_step'], 'layer', d['operator_layer']['ms_per_step'], 'model', d['model_step']['ms_per_step'], 'peak', d['model_step'].get('peak_mem_GB'))"
