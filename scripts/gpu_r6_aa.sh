#!/bin/bash
# round 6: HyenaDNALM padding ONE long odd-length sequence to a multiple of 64 (HYENA_LM_PAD_SINGLE_MIN): model step at the trainer's B = 1 lengths
out=gpurun_out/r6aa; mkdir -p $out
for L in 1048575 999999 449999; do
  for m in 8192 100000000; do
    echo "== L $L HYENA_LM_PAD_SINGLE_MIN=$m" >> $out/model_single_pad.txt
    HYENA_LM_PAD_SINGLE_MIN=$m python scripts/bench_model.py $L 1 256 6 2>&1 | tail -1 | cut -c1-200 >> $out/model_single_pad.txt
  done
done
echo "== L 1048576 (aligned)" >> $out/model_single_pad.txt
python scripts/bench_model.py 1048576 1 256 6 2>&1 | tail -1 | cut -c1-200 >> $out/model_single_pad.txt
cat $out/model_single_pad.txt
