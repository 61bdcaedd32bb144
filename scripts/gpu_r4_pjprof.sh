#!/bin/bash
# round 4: the MLP kernels: parity, phase times (profiling build), speed, with and without the global stores
TAG=${1:-r4pj}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_proj.py tests/test_gpu_block.py -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest.txt
HYENA_FFTCONV_LIB=$R/build/libhyena_pjprof.so timeout 300 python scripts/pj_phase_profile.py 2>&1 | grep -v amdgpu.ids | tee $OUT/mlp_phases.txt
timeout 300 python scripts/bench_proj.py "1048576 1 256" "32768 8 256" "131072 2 128" 2>&1 | grep -v amdgpu.ids | tee $OUT/bench_proj.txt
echo "== without the global stores" | tee -a $OUT/bench_proj.txt
HYENA_FFTCONV_LIB=$R/build/libhyena_nost.so timeout 300 python scripts/bench_proj.py "1048576 1 256" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/bench_proj.txt
