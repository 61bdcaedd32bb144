#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r2b; mkdir -p $OUT
cd $R
( timeout 300 ./build/xcd_flags ) > $OUT/xcd_flags.txt 2>&1
timeout 300 python scripts/oc_times.py "1024 8 128" "4096 8 256" "16384 8 256" "32768 8 256" "32768 1 256" > $OUT/oc_times.txt 2>&1
timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $OUT/pytest_parity.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest_parity.txt
cat $OUT/oc_times.txt; tail -5 $OUT/pytest_parity.txt; cat $OUT/xcd_flags.txt
