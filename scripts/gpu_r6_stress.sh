#!/bin/bash
# round 6: randomised double-run stress of the long convolution (incl. du + dk in one launch at M = 16384), the operator (out_proj generation 2 by default;
# a pass with in_proj generation 2, the dgrad kernel and the add + LayerNorm epilogue forced) and the 16-bit filter, on the final tree
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6stress; mkdir -p $OUT
cd $R
timeout 260 python scripts/gpu_stress_parity.py 170 21 2>&1 | tail -3 | tee $OUT/stress_parity.txt
HYENA_FFTCONV_DUDK=1 timeout 200 python scripts/gpu_stress_parity.py 110 22 2>&1 | tail -3 | tee $OUT/stress_parity_dudk.txt
timeout 260 python scripts/gpu_stress_operator.py 170 21 2>&1 | tail -3 | tee $OUT/stress_operator.txt
HYENA_INPROJ_KERNEL=2 HYENA_OUTPROJ_DGRAD_MFMA=1 timeout 200 python scripts/gpu_stress_operator.py 110 22 2>&1 | tail -3 | tee $OUT/stress_operator_gen2.txt
HYENA_OUTPROJ_KERNEL=1 timeout 160 python scripts/gpu_stress_operator.py 80 23 2>&1 | tail -3 | tee $OUT/stress_operator_outproj_gen1.txt
timeout 200 python scripts/gpu_stress_filter16.py 110 2>&1 | tail -3 | tee $OUT/stress_filter16.txt
STRESS_ORDERS=3,4,3,5 timeout 260 python scripts/gpu_stress_operator.py 170 24 2>&1 | tail -3 | tee $OUT/stress_operator_orders.txt
