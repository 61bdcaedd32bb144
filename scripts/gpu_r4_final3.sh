#!/bin/bash
# round 4, final record of the tree: whole GPU suite + smoke, operator stress, default bench line (with sweep), kernel stats of the headline and of the model step
TAG=${1:-r4f}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
bash scripts/gpu_tests.sh ${TAG}_tests
timeout 200 python scripts/gpu_stress_operator.py 60 7 2>&1 | tail -3 | tee $OUT/stress_operator.txt
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 1200 $OUT/bench_default.json
bash scripts/gpu_prof_bench.sh ${TAG}_prof1m --no-operator --no-model --no-sweep | tail -12 | cut -c1-150
bash scripts/gpu_prof_model.sh ${TAG}_model 1048576 1 256 | head -8 | cut -c1-150
