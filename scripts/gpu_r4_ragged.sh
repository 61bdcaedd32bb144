#!/bin/bash
# round 4: the fused out_proj kernel at ragged lengths (pulled-back last tile): parity tests, the operator stress, speed next to the aligned lengths
TAG=${1:-r4rag}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_proj.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest_proj.txt
timeout 200 python scripts/gpu_stress_operator.py 100 5 2>&1 | tail -4 | tee $OUT/stress_operator.txt
timeout 300 python scripts/bench_outproj.py "1048576 1 256" "1048575 1 256" "32768 8 256" "32767 8 256" "65535 2 128" "65536 2 128" 2>&1 | tee $OUT/bench_outproj.txt
