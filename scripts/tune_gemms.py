"""Round 6 experiment: PyTorch TunableOp over the library GEMMs of the model step (19 % of it): python scripts/tune_gemms.py L B D n_layer out.csv
1. untuned step time; 2. two steps with tuning on (results -> out.csv); 3. step time with the tuned solutions (tuning off).
Result (profiles/r6s_tunableop.txt): five mm shapes tuned (the slice-batched bmm(out_dtype=fp32) weight gradients are not TunableOp ops), step 40.59 -> 40.79 ms at
2 layers x 2^20: the library's own heuristic already picks within noise of the best solution -- not adopted."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.cuda.tunable as tn  # noqa: E402
import bench  # noqa: E402

L, B, D, n_layer = (int(x) for x in sys.argv[1:5])
out = sys.argv[5]
dev = torch.device("cuda", 0)


def step(steps=6):
    r = bench.model_step(L, D, B, torch.bfloat16, dev, n_layer=n_layer, steps=steps, graphed_ok=False)
    return r["min_ms"], r["median_ms"]


print("untuned", step(), flush=True)
tn.enable(True)
tn.tuning_enable(True)
tn.set_max_tuning_duration(int(os.environ.get("TUNE_MS", "150")))
tn.set_max_tuning_iterations(int(os.environ.get("TUNE_ITERS", "20")))
tn.set_filename(out)
print("tuning", step(2), flush=True)
tn.tuning_enable(False)
print("tuned", step(), flush=True)
print("tuned again", step(), flush=True)
for r in tn.get_results():
    print(r)
