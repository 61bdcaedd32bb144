#!/bin/bash
# round 4, call F: exchange 1 as complex halves (dk / one-launch kernels only = regular; everywhere = x1all; nowhere = x1none) + the
# stale-data probe for the L2-resident exchange + parity tests of the workspace-free plan
TAG=${1:-r4f}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
CFGS='"32768 8 256" "16384 8 256" "8192 8 256" "4096 16 256" "32768 2 256" "2048 64 128" "1024 64 128"'
for v in regular x1none x1all regular x1none; do
  echo "== $v" | tee -a $OUT/ab.txt
  if [ $v = regular ]; then unset HYENA_FFTCONV_LIB; else export HYENA_FFTCONV_LIB=$R/build/libhyena_$v.so; fi
  eval timeout 300 python scripts/oc_times.py $CFGS 2>&1 | grep "L=" | tee -a $OUT/ab.txt
  timeout 200 python bench.py --seq-len 1024 --batch 8 --d-model 128 --steps 200 --warmup 20 --no-cpu-baseline --no-operator --no-model --no-sweep 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench 1024x8x128: %.2f us/step frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['frac']))" | tee -a $OUT/ab.txt
done
unset HYENA_FFTCONV_LIB
timeout 120 ./build/xcd_stale_probe 2>&1 | tee $OUT/stale_probe.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small.py tests/test_gpu_binding.py tests/test_gpu_proj.py -m gpu -x -q 2>&1 | tail -5 | tee $OUT/pytest_gpu.txt
