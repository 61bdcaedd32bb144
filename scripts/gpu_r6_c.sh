#!/bin/bash
# round 6, call C: generation 2 of the out_proj and in_proj kernels: the -m gpu tests of both generations, then A/B timings
TAG=${1:-r6c}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_proj.py -m gpu -q -x -k "outproj or out_proj or inproj or mfma_projection" 2>&1 | tail -8 | tee $OUT/pytest_proj.txt
timeout 600 python scripts/bench_outproj.py "1048576 1 256" "1048575 1 256" "32768 8 256" "32767 8 256" "160000 2 256" "1024 256 128" "1023 256 128" 2>&1 | tee $OUT/bench_outproj.txt
timeout 600 python scripts/bench_inproj.py "1048576 1 256" "1048575 1 256" "32768 8 256" "32767 8 256" "160000 2 256" "1024 256 128" "1023 256 128" 2>&1 | tee $OUT/bench_inproj.txt
