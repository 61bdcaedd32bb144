#!/bin/bash
TAG=${1:-r6l}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_proj.py -m gpu -q -x -k "add_norm" 2>&1 | tail -3 | tee $OUT/pytest_ln.txt
GEN=2 timeout 600 python scripts/bench_outproj.py "1048576 1 256" "1048575 1 256" "32768 8 256" "160000 2 256" "1024 256 128" "1023 256 128" 2>&1 | grep -v amdgpu | tee $OUT/bench_outproj.txt
