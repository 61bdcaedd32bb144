#!/bin/bash
# round 5: out_proj with the block's add + LayerNorm in its epilogue: micro-benchmark, then the model step with and without it
TAG=${1:-r5c}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 300 python scripts/bench_outproj.py "1048576 1 256" "1048575 1 256" "32768 8 256" "131072 2 128" 2>&1 | grep -v amdgpu.ids | tee $OUT/bench_outproj.txt
for knob in 1 0; do
  HYENA_ADD_NORM_FUSED=$knob timeout 600 python scripts/bench_model.py 1048576 1 256 8 2>&1 | tail -1 | cut -c1-200 | tee -a $OUT/model_ab.txt
done
