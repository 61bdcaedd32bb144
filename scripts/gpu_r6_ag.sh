#!/bin/bash
out=gpurun_out/r6ag; mkdir -p $out
python -m pytest tests/test_gpu_filter_stream.py -q -x 2>&1 | tail -5 | tee $out/pytest_stream.txt
for cfg in "1048576 1 256 8 8" "32768 8 256 8 8" "159999 2 256 8 8" "1023 256 128 12 2" "32768 8 128 8 4"; do
  for sd in 0 auto; do
    echo "== $cfg HYENA_FILTER_SIDE_STREAM=$sd" >> $out/filter_side.txt
    HYENA_FILTER_SIDE_STREAM=$sd python scripts/bench_model.py $cfg 2>&1 | tail -1 | cut -c1-140 >> $out/filter_side.txt
  done
done
cat $out/filter_side.txt
