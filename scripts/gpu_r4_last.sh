#!/bin/bash
# round 4, last call: randomised double-run stress of the final kernels (operator incl. the fused out_proj; long convolution incl. dk with its
# new barrier), the whole GPU suite + smoke, the default bench line
TAG=${1:-r4zz}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 200 python scripts/gpu_stress_operator.py 90 4 2>&1 | tail -4 | tee $OUT/stress_operator.txt
timeout 240 python scripts/gpu_stress_parity.py 120 4 2>&1 | tail -4 | tee $OUT/stress_parity.txt
bash scripts/gpu_tests.sh ${TAG}_tests
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 600 $OUT/bench_default.json
