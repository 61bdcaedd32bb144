#!/bin/bash
# round 4 final record: whole -m gpu suite + smoke, the default bench line (with `sweep`), rocprofv3 kernel stats of the bench at the
# contract configurations, and of the model step
TAG=${1:-r4z}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
bash scripts/gpu_tests.sh ${TAG}_tests
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 1500 $OUT/bench_default.json
bash scripts/gpu_prof_bench.sh ${TAG}_prof1m --no-operator --no-model --no-sweep
bash scripts/gpu_prof_bench.sh ${TAG}_prof32k --seq-len 32768 --batch 8 --no-operator --no-model --no-sweep
bash scripts/gpu_prof_bench.sh ${TAG}_prof160k --seq-len 160000 --batch 2 --no-operator --no-model --no-sweep
bash scripts/gpu_prof_bench.sh ${TAG}_prof450k --seq-len 450560 --batch 1 --no-operator --no-model --no-sweep
bash scripts/gpu_prof_model.sh ${TAG}_model 1048576 1 256 | head -30 | cut -c1-150
