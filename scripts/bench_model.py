"""The full hyenadna model step alone (bench.model_step), for profiling: python scripts/bench_model.py L B D [steps] [n_layer]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

L, B, D = (int(x) for x in sys.argv[1:4])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
n_layer = int(sys.argv[5]) if len(sys.argv) > 5 else 8
r = bench.model_step(L, D, B, torch.bfloat16, torch.device("cuda", 0), n_layer=n_layer, steps=steps, graphed_ok=False)
print(r)
