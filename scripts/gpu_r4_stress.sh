#!/bin/bash
# round 4, last: randomised double-run stress of the long convolution (after the row0_bwd change) and of the operator (after the MFMA kernels' rework)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4stress; mkdir -p $OUT
cd $R
timeout 200 python scripts/gpu_stress_parity.py 110 9 2>&1 | tail -3 | tee $OUT/stress_parity.txt
timeout 200 python scripts/gpu_stress_operator.py 110 9 2>&1 | tail -3 | tee $OUT/stress_operator.txt
