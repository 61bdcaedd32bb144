#!/bin/bash
# the N > 1 leg of bench.py on ONE GPU (two ranks sharing cuda:0, gloo): exercises the DDP-wrapped model step and the side stream from two processes
TAG=${1:-r4ddp}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --share-gpu0 \
    --steps 5 --warmup 2 --seq-len 160000 --batch 1 --model-layers 2 2>&1 | grep -E '^\{|Error|error' | cut -c1-1500 | tee $OUT/bench_2ranks.txt
