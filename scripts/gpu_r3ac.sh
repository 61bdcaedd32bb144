#!/bin/bash
# dk at B = 1 through spectrum + conjugate convolution: timings with and without (HYENA_FFTCONV_DK1=0), then the whole GPU suite + smoke
TAG=${1:-r3ac}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
echo "== dk at B = 1: spectrum of u + convolution with the conjugate" | tee $OUT/dk1.txt
timeout 200 python scripts/oc_times.py "32768 1 256" "30000 1 256" "16384 1 256" "8192 1 256" "4096 1 256" 2>&1 | grep "L=" | tee -a $OUT/dk1.txt
echo "== HYENA_FFTCONV_DK1=0: dk_kernel" | tee -a $OUT/dk1.txt
HYENA_FFTCONV_DK1=0 timeout 200 python scripts/oc_times.py "32768 1 256" "30000 1 256" "16384 1 256" "8192 1 256" "4096 1 256" 2>&1 | grep "L=" | tee -a $OUT/dk1.txt
bash scripts/gpu_tests.sh $TAG
