#!/bin/bash
# round 6: the stabilised tiny-LM training test over several seeds + the one-step gradient test
out=gpurun_out/${1:-r6r}; mkdir -p $out
python -m pytest tests/test_gpu_block.py -q -x 2>&1 | tail -15 > $out/pytest_block.txt
python - > $out/tiny_lm_seeds.txt 2>&1 <<'PY'
import torch
from tests._tiny_lm import train
for seed in range(6):
    for lr, wu in ((1.5e-3, 8), (3e-3, 0)):
        l = train("cuda", steps=40, d=128, L=2048, B=4, n_layer=2, autocast_dtype=torch.bfloat16, seed=seed, lr=lr, warmup=wu)
        print(seed, lr, wu, "first %.3f last %.3f max_tail %.3f ratio %.3f" % (l[0], l[-1], max(l[-5:]), l[-1] / l[0]), flush=True)
PY
cat $out/pytest_block.txt; cat $out/tiny_lm_seeds.txt
