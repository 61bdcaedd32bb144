#!/bin/bash
# round 4: in_proj kernel with LDS-direct operand loads: parity + speed
TAG=${1:-r4ip}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_proj.py tests/test_gpu_cm.py tests/test_gpu_block.py tests/test_gpu_contract.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/pytest.txt
timeout 300 python scripts/bench_proj.py "1048576 1 256" "32768 8 256" "131072 2 128" 2>&1 | grep -v amdgpu.ids | tee $OUT/bench_proj.txt
