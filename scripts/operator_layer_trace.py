"""Per-step event times of bench.operator_layer (diagnostic): python scripts/operator_layer_trace.py L B D steps"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

L, B, D, steps = (int(x) for x in sys.argv[1:5])
dev = torch.device("cuda", 0)
orig = bench._timed_steps


def traced(step, steps_, dev_):
    cpu = []

    def timed_step():
        t0 = time.perf_counter()
        step()
        cpu.append(round((time.perf_counter() - t0) * 1e3, 1))
    ts = orig(timed_step, steps_, dev_)
    st = torch.cuda.memory_stats(dev_)
    print("event ms", [round(t, 1) for t in ts])
    print("cpu issue ms", cpu)
    print("alloc_retries", st.get("num_alloc_retries"), "reserved GiB", round(torch.cuda.memory_reserved(dev_) / 2 ** 30, 1), flush=True)
    return ts


bench._timed_steps = traced
for rep in range(2):
    r = bench.operator_layer(L, D, B, torch.bfloat16, dev, steps=steps)
    print({k: round(r[k], 2) for k in ("ms_per_step", "median_ms", "min_ms")})
