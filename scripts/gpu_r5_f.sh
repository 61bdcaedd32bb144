#!/bin/bash
TAG=${1:-r5f}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_proj.py tests/test_gpu_block.py tests/test_gpu_runner.py -x -q -m gpu 2>&1 | tail -8 | tee $OUT/pytest.txt
for cfg in "32767 8" "32768 8" "159999 2" "160000 2" "449999 1" "450000 1"; do
  timeout 300 python scripts/bench_operator.py $cfg fused 2>&1 | tail -1 | tee -a $OUT/op.txt
done
