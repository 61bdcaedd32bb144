#!/bin/bash
# One gpurun call: smoke + GPU parity tests + bench lines + rocprofv3 kernel stats.  Everything lands in gpurun_out/.
# usage: gpurun --timeout 1500 -- 'bash scripts/gpu_check.sh [tag]'
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== smoke"; timeout 600 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.log
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
echo "== bench L=1M"; timeout 600 python bench.py > $OUT/bench_1m.json 2> $OUT/bench_1m.err; echo "bench rc=$?"; cat $OUT/bench_1m.json
for cfg in "1024 128 8" "32768 256 8" "160000 256 2" "450560 256 1"; do
  set -- $cfg
  echo "== bench L=$1 d=$2 B=$3"
  timeout 300 python bench.py --seq-len $1 --d-model $2 --batch $3 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_L$1.json 2> $OUT/bench_L$1.err
  cat $OUT/bench_L$1.json
done
echo "== bench L=1M fwd only / chunk sweep"
for c in 2 4 8 16 32; do
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --chunk $c 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('chunk', r['config']['chunk'], 'ms', round(r['ms_per_step'],3), 'frac', round(r['roofline']['frac'],4))"
done
echo "== rocprofv3 kernel stats"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench1m -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/rocprof.log 2>&1
echo "rocprof rc=$?"
find $OUT/prof -name '*kernel_stats*' | head -3
f=$(find $OUT/prof -name '*kernel_stats*.csv' | head -1); [ -n "$f" ] && head -20 "$f"
# keep the merge small: drop the raw trace, keep stats
find $OUT/prof -name '*kernel_trace*' -size +20M -delete
