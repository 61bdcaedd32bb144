"""PyTorch TunableOp over the library GEMMs that remain on the model step (out_proj, fc2, input / weight gradients, lm_head):
one tuning pass at the given configurations, results written to <out>; then the same steps timed with the default heuristic and
with the tuned solutions.  usage: python scripts/tune_gemms.py <out.csv> "L B D" ...   (bf16 autocast, 2 layers are enough:
every layer has the same shapes)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.cuda.tunable as tunable  # noqa: E402

import bench  # noqa: E402

out = sys.argv[1]
cfgs = [tuple(int(x) for x in c.split()) for c in sys.argv[2:]]
dev = torch.device("cuda", 0)


def run(tag):
    for L, B, D in cfgs:
        r = bench.model_step(L, D, B, torch.bfloat16, dev, n_layer=2, steps=5, warmup=2, graphed_ok=False)
        print(f"{tag}: L={L} B={B} D={D}: model step (2 layers) {r['ms_per_step']:.3f} ms", flush=True)


run("default heuristic")
tunable.enable(True)
tunable.tuning_enable(True)
tunable.set_filename(out)
tunable.set_max_tuning_duration(15)
tunable.set_max_tuning_iterations(20)
t0 = time.time()
run("tuning pass")
print(f"tuning took {time.time() - t0:.1f} s; {len(tunable.get_results())} entries", flush=True)
tunable.tuning_enable(False)
run("tuned")
tunable.write_file = getattr(tunable, "write_file", None)
try:
    torch.cuda.tunable.write_file(out)
except Exception as e:  # noqa: BLE001
    print("write_file:", e)
