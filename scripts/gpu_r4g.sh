#!/bin/bash
# round 4, call G: L2-resident exchange without an invalidate -- the probe's rate table next to round 3's experiment on the same box
TAG=${1:-r4g}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 200 ./build/xcd_stale_probe 2>&1 | tee $OUT/stale_probe.txt
echo "==== scripts/xcd_flags.hip (round 3's experiment) on the same box" | tee -a $OUT/stale_probe.txt
timeout 300 ./build/xcd_flags 2>&1 | head -60 | tee -a $OUT/stale_probe.txt
