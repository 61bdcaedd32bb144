#!/bin/bash
# round 5: out_proj's input gradient with the gate backward in its epilogue: GPU tests, micro-benchmark, the operator layer and the model step with and without it
TAG=${1:-r5d}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_proj.py -x -q -m gpu 2>&1 | tail -5 | tee $OUT/pytest_proj.txt
timeout 300 python scripts/bench_dgrad.py "1048576 1 256" "1048575 1 256" "32768 8 256" "32767 8 256" "160000 2 256" "131072 2 128" 2>&1 | grep -v amdgpu.ids | tee $OUT/bench_dgrad.txt
for knob in 1 0; do
  HYENA_OUTPROJ_DGRAD_MFMA=$knob timeout 600 python scripts/bench_operator.py 1048576 1 fused 2>&1 | tail -1 | tee -a $OUT/op_ab.txt
  HYENA_OUTPROJ_DGRAD_MFMA=$knob timeout 600 python scripts/bench_model.py 1048576 1 256 8 2>&1 | tail -1 | cut -c1-200 | tee -a $OUT/model_ab.txt
done
