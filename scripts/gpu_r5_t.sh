#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r5t; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_proj.py tests/test_gpu_contract.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest.txt
for L in 1048576 1048575; do
  timeout 600 python scripts/bench_model.py $L 1 256 8 2>&1 | tail -1 | cut -c1-160 | tee -a $OUT/model.txt
  timeout 300 python scripts/bench_operator.py $L 1 fused 2>&1 | tail -1 | tee -a $OUT/op.txt
done
