#!/bin/bash
# Fast iteration on the GPU box: parity tests, L=1M bench with a chunk sweep, rocprofv3 kernel stats.
# usage: gpurun --timeout 900 -- 'bash scripts/gpu_iter.sh <tag> [chunks...]'
TAG=${1:-it}; shift
CHUNKS=${@:-"0"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 600 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
for c in $CHUNKS; do
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --chunk $c 2>$OUT/bench_c$c.err | tee $OUT/bench_c$c.json | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('L=1M chunk', r['config']['chunk'], 'ms', round(r['ms_per_step'],3), 'frac', round(r['roofline']['frac'],4), 'Mnt/s', round(r['value']/1e6,1))"
done
for cfg in "32768 256 8" "160000 256 2"; do
  set -- $cfg
  timeout 300 python bench.py --seq-len $1 --d-model $2 --batch $3 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('L', r['config']['seq_len'], 'B', r['config']['batch_per_gpu'], 'chunk', r['config']['chunk'], 'ms', round(r['ms_per_step'],3), 'frac', round(r['roofline']['frac'],4), 'Mnt/s', round(r['value']/1e6,1))"
done
echo "== rocprofv3 kernel stats (L=1M, default chunk)"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench1m -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/rocprof.log 2>&1
echo "rocprof rc=$?"
python $R/scripts/rocpd_stats.py $(find $OUT/prof -name '*.db' | head -1) $OUT/kernel_stats.csv | grep hyena | cut -c1-150
