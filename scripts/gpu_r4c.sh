#!/bin/bash
# round 4, call C: twiddle tables in LDS -- A/B against the previous kernels (build/libhyena_old.so), phase timeline, parity tests
TAG=${1:-r4c}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
CFGS='"32768 8 256" "16384 8 256" "8192 8 256" "4096 16 256" "32768 2 256" "2048 64 128" "32768 1 256" "1024 64 128"'
for v in new old new old; do
  echo "== $v" | tee -a $OUT/ab.txt
  if [ $v = old ]; then export HYENA_FFTCONV_LIB=$R/build/libhyena_old.so; else unset HYENA_FFTCONV_LIB; fi
  eval timeout 300 python scripts/oc_times.py $CFGS 2>&1 | grep "L=" | tee -a $OUT/ab.txt
  timeout 200 python bench.py --seq-len 1024 --batch 8 --d-model 128 --steps 200 --warmup 20 --no-cpu-baseline --no-operator --no-model --no-sweep 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench 1024x8x128: %.2f us/step frac %.3f graph %s' % (d['ms_per_step']*1e3, d['roofline']['frac'], d['config']['hipgraph_replay']))" | tee -a $OUT/ab.txt
done
unset HYENA_FFTCONV_LIB
HYENA_FFTCONV_LIB=$R/build/libhyena_prof.so timeout 300 python scripts/oc_phase_profile.py "32768 8 256" "16384 8 256" 2>&1 | tee $OUT/phases.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small.py tests/test_gpu_contract.py tests/test_gpu_binding.py -m gpu -x -q 2>&1 | tail -5 | tee $OUT/pytest_gpu.txt
