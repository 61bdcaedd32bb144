#!/bin/bash
# round 6: column sums of dx0 inside add_norm_bwd (the side table of _gradsum) -- tests + model step A/B
out=gpurun_out/r6ae; mkdir -p $out
python -m pytest tests/test_gpu_block.py tests/test_gpu_contract.py -q -x 2>&1 | tail -3 | tee $out/pytest.txt
for L in "1048576 1 256 8 8" "1023 256 128 12 2" "32768 8 256 8 8"; do
  for gs in 1 0; do
    echo "== $L HYENA_GRADSUM=$gs" >> $out/model_gradsum.txt
    HYENA_GRADSUM=$gs python scripts/bench_model.py $L 2>&1 | tail -1 | cut -c1-150 >> $out/model_gradsum.txt
  done
done
cat $out/model_gradsum.txt
