#!/bin/bash
# round 6: software-pipelined operand loads in the 16-bit filter's backward kernels (F16_PIPE) against the load / wait / compute rounds (build/libhyena_nopipe.so)
out=gpurun_out/r6ac; mkdir -p $out
for rep in 1 2; do
for L in 1048576 1048575 159999 32768; do
  echo "== pipe L $L" >> $out/filter16_pipe.txt
  python scripts/bench_filter.py $L 256 --fused16-only 2>&1 | tail -1 >> $out/filter16_pipe.txt
  echo "== no pipe L $L" >> $out/filter16_pipe.txt
  HYENA_FFTCONV_LIB=$PWD/build/libhyena_nopipe.so python scripts/bench_filter.py $L 256 --fused16-only 2>&1 | tail -1 >> $out/filter16_pipe.txt
done
done
echo "== L 1048576 d_model 128" >> $out/filter16_pipe.txt
python scripts/bench_filter.py 1048576 128 --fused16-only 2>&1 | tail -1 >> $out/filter16_pipe.txt
HYENA_FFTCONV_LIB=$PWD/build/libhyena_nopipe.so python scripts/bench_filter.py 1048576 128 --fused16-only 2>&1 | tail -1 >> $out/filter16_pipe.txt
cat $out/filter16_pipe.txt
python -m pytest tests/test_gpu_filter.py -q 2>&1 | tail -2
