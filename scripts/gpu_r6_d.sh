#!/bin/bash
# round 6, call D: in_proj generation 2 in both workgroup shapes
TAG=${1:-r6d}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_proj.py -m gpu -q -x -k "inproj or mfma_projection" 2>&1 | tail -4 | tee $OUT/pytest_inproj.txt
timeout 600 python scripts/bench_inproj.py "1048576 1 256" "1048575 1 256" "32768 8 256" "32767 8 256" "160000 2 256" "450560 1 256" "1024 256 128" "1023 256 128" 2>&1 | tee $OUT/bench_inproj.txt
