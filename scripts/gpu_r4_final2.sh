#!/bin/bash
# round 4, after the MFMA kernels' rework: whole GPU suite + smoke, operator stress, the default bench line
TAG=${1:-r4y}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
bash scripts/gpu_tests.sh ${TAG}_tests
timeout 200 python scripts/gpu_stress_operator.py 60 6 2>&1 | tail -3 | tee $OUT/stress_operator.txt
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 1500 $OUT/bench_default.json
