#!/bin/bash
# the whole -m gpu suite + smoke, logs under gpurun_out/<tag>
TAG=${1:-tests}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 2400 python -m pytest tests -q -m gpu --durations=12 > $OUT/pytest_gpu.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.txt
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.txt 2>&1
echo "smoke exit $?" >> $OUT/smoke.txt
tail -8 $OUT/pytest_gpu.txt; tail -3 $OUT/smoke.txt
