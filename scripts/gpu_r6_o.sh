#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6o; mkdir -p $OUT
cd $R
timeout 600 python scripts/bench_dgrad.py "1024 256 128" "1023 256 128" "32768 8 256" "32767 8 256" "160000 2 256" "159999 2 256" "1048576 1 256" "4096 64 256" 2>&1 | grep -v amdgpu | tee $OUT/bench_dgrad.txt
