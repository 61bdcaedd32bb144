#!/bin/bash
# round 5: the dgrad kernel compiled for 3 (product) and 2 (build/libhyena_dg2.so) workgroups per CU
TAG=${1:-r5e}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_proj.py -x -q -m gpu 2>&1 | tail -5 | tee $OUT/pytest_proj.txt
echo "--- DG_WGS = 3 (product)" | tee $OUT/bench_dgrad.txt
timeout 300 python scripts/bench_dgrad.py "1048576 1 256" "1048575 1 256" "32768 8 256" "32767 8 256" "160000 2 256" "159999 2 256" "450560 1 256" "131072 2 128" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/bench_dgrad.txt
echo "--- DG_WGS = 2" | tee -a $OUT/bench_dgrad.txt
HYENA_FFTCONV_LIB=$R/build/libhyena_dg2.so timeout 300 python scripts/bench_dgrad.py "1048576 1 256" "32768 8 256" "160000 2 256" "131072 2 128" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/bench_dgrad.txt
for knob in 1 0; do
  HYENA_OUTPROJ_DGRAD_MFMA=$knob timeout 600 python scripts/bench_model.py 1048576 1 256 8 2>&1 | tail -1 | cut -c1-200 | tee -a $OUT/model_ab.txt
done
