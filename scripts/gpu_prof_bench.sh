#!/bin/bash
# rocprofv3 kernel stats of bench.py with arbitrary arguments: bash scripts/gpu_prof_bench.sh <tag> <bench args...>
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof -o b -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $OUT/log.txt 2>&1
grep -h "^{\"metric\"" $OUT/log.txt | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('ms', r['ms_per_step'], 'frac', r['roofline']['frac'])"
python $R/scripts/rocpd_stats.py $(find $OUT/prof -name '*.db' | head -1) $OUT/stats.csv | grep hyena | cut -c1-150
find $OUT/prof -name '*.db' -size +30M -delete
