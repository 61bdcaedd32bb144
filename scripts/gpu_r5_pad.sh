#!/bin/bash
# round 5: B > 1 sequences of odd length padded inside in_proj: the layer at the real shapes with (default) and without the padding, then the contract tests
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r5pad; mkdir -p $OUT
for knob in 1 0; do
  for cfg in "32767 8" "159999 2"; do
    HYENA_PAD_SEQUENCES=$knob timeout 300 python scripts/bench_operator.py $cfg fused 2>&1 | tail -1 | sed "s/^/pad=$knob /" | tee -a $OUT/op.txt
  done
done
for cfg in "32768 8" "160000 2"; do timeout 300 python scripts/bench_operator.py $cfg fused 2>&1 | tail -1 | tee -a $OUT/op.txt; done
timeout 900 python -m pytest tests/test_gpu_contract.py tests/test_gpu_proj.py tests/test_gpu_runner.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/pytest.txt
