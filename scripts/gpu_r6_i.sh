#!/bin/bash
# round 6, call I: du + dk from one transform of dout (config-2 experiment); the fftconv parity tests with the knob on
TAG=${1:-r6i}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 600 python scripts/bench_dudk.py 2>&1 | tee $OUT/bench_dudk.txt
HYENA_FFTCONV_DUDK=1 timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "random or onchip or workspace_free or properties or contract_configs" 2>&1 | tail -4 | tee $OUT/pytest_dudk.txt
