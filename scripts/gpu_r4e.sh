#!/bin/bash
# round 4, call E: fused out_proj kernel -- kernel timings (regular = 4 staging batches, nb2 = 2 batches with spills), layer / model A/B, GPU tests
TAG=${1:-r4e}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for v in regular nb2; do
  echo "== $v" | tee -a $OUT/outproj.txt
  if [ $v = nb2 ]; then export HYENA_FFTCONV_LIB=$R/build/libhyena_nb2.so; else unset HYENA_FFTCONV_LIB; fi
  timeout 300 python scripts/bench_outproj.py "1048576 1 256" "32768 8 256" "160000 2 256" "65536 2 128" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/outproj.txt
done
unset HYENA_FFTCONV_LIB
for on in 1 0 1 0; do
  HYENA_OUTPROJ_MFMA=$on timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sweep 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('OUTPROJ_MFMA=$on: conv %.3f ms, operator_layer %.3f ms, model_step %.2f ms (peak %.1f GB)' % (d['ms_per_step'], d['operator_layer']['ms_per_step'], d['model_step']['ms_per_step'], d['model_step']['peak_mem_GB']))" | tee -a $OUT/outproj.txt
done
timeout 900 python -m pytest tests/test_gpu_contract.py tests/test_gpu_proj.py tests/test_gpu_block.py tests/test_gpu_binding.py tests/test_gpu_runner.py -m gpu -x -q 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
