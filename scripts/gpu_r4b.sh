#!/bin/bash
# round 4, call B: in-kernel phase timeline of conv_kernel (profiling build), 32768 / 16384 / 8192
TAG=${1:-r4b}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
HYENA_FFTCONV_LIB=$R/build/libhyena_prof.so timeout 300 python scripts/oc_phase_profile.py "32768 8 256" "32768 1 256" "16384 8 256" "8192 8 256" 2>&1 | tee $OUT/phases.txt
echo "== regular build timings" | tee -a $OUT/phases.txt
timeout 300 python scripts/oc_times.py "32768 8 256" "16384 8 256" 2>&1 | grep "L=" | tee -a $OUT/phases.txt
echo "== profiling build timings" | tee -a $OUT/phases.txt
HYENA_FFTCONV_LIB=$R/build/libhyena_prof.so timeout 300 python scripts/oc_times.py "32768 8 256" "16384 8 256" 2>&1 | grep "L=" | tee -a $OUT/phases.txt
