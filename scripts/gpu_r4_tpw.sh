#!/bin/bash
# experiment: run length (tiles per workgroup) of the in_proj kernel vs its store pattern
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4tpw; mkdir -p $OUT
cd $R
for t in 0 8 16 24 31 32 33 48 64 96 128; do
  if [ $t = 0 ]; then echo "== default" | tee -a $OUT/tpw.txt; HYENA_FFTCONV_LIB=$R/build/libhyena_tpw.so timeout 100 python scripts/bench_proj.py "1048576 1 256" 2>&1 | grep "fused MFMA" | tee -a $OUT/tpw.txt
  else echo "== tiles_per_wg $t" | tee -a $OUT/tpw.txt; HYENA_PJ_TPW=$t HYENA_FFTCONV_LIB=$R/build/libhyena_tpw.so timeout 100 python scripts/bench_proj.py "1048576 1 256" 2>&1 | grep "fused MFMA" | tee -a $OUT/tpw.txt; fi
done
