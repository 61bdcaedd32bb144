#!/bin/bash
TAG=${1:-bench}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 900 python bench.py > $OUT/bench_1m.json 2> $OUT/bench_1m.err
echo "bench exit $?" >> $OUT/bench_1m.err
timeout 600 python bench.py --seq-len 32768 --batch 8 --no-cpu-baseline > $OUT/bench_32k.json 2> $OUT/bench_32k.err
timeout 600 python bench.py --seq-len 1024 --batch 8 --d-model 128 --no-cpu-baseline > $OUT/bench_1k.json 2> $OUT/bench_1k.err
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "contract" > $OUT/pytest_contract.txt 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/%s/bench_*.json" % "TAGX".replace("TAGX", __import__("os").environ.get("TAG", "")))):
    pass
PY
for f in $OUT/bench_*.json; do python -c "
import json,sys
r=json.loads(open('$f').read().strip().splitlines()[-1])
print('$f', 'ms %.4f'%r['ms_per_step'], 'frac %.3f'%r['roofline']['frac'], 'valu %.3f'%r['roofline_valu']['frac'], 'op', r.get('operator_layer',{}).get('ms_per_step'), 'model', r.get('model_step'))
"; done
tail -3 $OUT/pytest_contract.txt; tail -3 $OUT/bench_1m.err
