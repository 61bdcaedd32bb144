#!/bin/bash
# Round 2, first GPU visit: micro-benchmarks, parity of the workspace-free plan on hardware, sweep, kernel stats.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r2a; mkdir -p $OUT
cd $R
( timeout 120 ./build/valu_rate ) > $OUT/valu_rate.txt 2>&1
( timeout 180 ./build/xcd_flags ) > $OUT/xcd_flags.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $OUT/pytest_parity.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest_parity.txt
: > $OUT/sweep.jsonl
for cfg in "1024 8 128" "32768 8 256" "32768 1 256" "16384 8 256" "160000 2 256" "1048576 1 256"; do
    set -- $cfg
    timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-operator --seq-len $1 --batch $2 --d-model $3 >> $OUT/sweep.jsonl 2>> $OUT/sweep.err
done
# the two-level plan at 32k for comparison
HYENA_FFTCONV_ONCHIP=0 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-operator --seq-len 32768 --batch 8 --d-model 256 >> $OUT/sweep_twolevel.jsonl 2>> $OUT/sweep.err
export TMPDIR=/tmp; cd /tmp
for cfg in "32768 8 256" "1024 8 128"; do
    set -- $cfg
    timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$1 -o b -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-operator --seq-len $1 --batch $2 --d-model $3 > $OUT/prof_$1.log 2>&1
    python $R/scripts/rocpd_stats.py $(find $OUT/prof_$1 -name '*.db' | head -1) $OUT/stats_$1.csv > /dev/null 2>&1
    find $OUT/prof_$1 -name '*.db' -size +20M -delete
done
cd $R
python - <<'PY'
import json
for f in ("gpurun_out/r2a/sweep.jsonl", "gpurun_out/r2a/sweep_twolevel.jsonl"):
    try:
        for l in open(f):
            a = json.loads(l); c = a["config"]
            print(f, c["seq_len"], c["batch_per_gpu"], c["channels"], "ms %.4f" % a["ms_per_step"], "frac %.3f" % a["roofline"]["frac"])
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $OUT/pytest_parity.txt
cat $OUT/valu_rate.txt; head -12 $OUT/xcd_flags.txt
