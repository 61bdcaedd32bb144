#!/bin/bash
# round 5: randomised double-run stress of the long convolution, the operator (pitched rows; dgrad kernel forced on in a second pass) and the 16-bit filter
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r5stress; mkdir -p $OUT
cd $R
timeout 260 python scripts/gpu_stress_parity.py 170 11 2>&1 | tail -3 | tee $OUT/stress_parity.txt
timeout 260 python scripts/gpu_stress_operator.py 170 11 2>&1 | tail -3 | tee $OUT/stress_operator.txt
HYENA_OUTPROJ_DGRAD_MFMA=1 timeout 200 python scripts/gpu_stress_operator.py 110 12 2>&1 | tail -3 | tee $OUT/stress_operator_dgrad.txt
timeout 200 python scripts/gpu_stress_filter16.py 110 2>&1 | tail -3 | tee $OUT/stress_filter16.txt
for cfg in "32767 8" "32768 8" "159999 2" "160000 2"; do
  timeout 300 python scripts/bench_operator.py $cfg fused 2>&1 | tail -1 | tee -a $OUT/op.txt
done
