#!/bin/bash
# round 6 experiment: the implicit filter (forward and, through autograd's stream rule, backward) on a second stream next to the projections
out=gpurun_out/r6af; mkdir -p $out
for cfg in "1048576 1 256 8 8" "32768 8 256 8 8" "159999 2 256 8 8"; do
  for sd in 0 1 0 1; do
    echo "== $cfg HYENA_FILTER_SIDE_STREAM=$sd" >> $out/filter_side.txt
    HYENA_FILTER_SIDE_STREAM=$sd python scripts/bench_model.py $cfg 2>&1 | tail -1 | cut -c1-140 >> $out/filter_side.txt
  done
done
cat $out/filter_side.txt
HYENA_FILTER_SIDE_STREAM=1 python -m pytest tests/test_gpu_contract.py -q -x -k "operator_at_contract or lm" 2>&1 | tail -2
