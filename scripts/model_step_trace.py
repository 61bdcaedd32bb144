"""Per-step times and allocator state of the model step (diagnostic): python scripts/model_step_trace.py L B D n_layer steps"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

L, B, D, n_layer, steps = (int(x) for x in sys.argv[1:6])
dev = torch.device("cuda", 0)
orig = bench._timed_steps


def traced(step, steps_, dev_):
    ts = orig(step, steps_, dev_)
    st = torch.cuda.memory_stats(dev_)
    print("times", [round(t, 1) for t in ts], "alloc_retries", st.get("num_alloc_retries"), "reserved GiB", round(torch.cuda.memory_reserved(dev_) / 2 ** 30, 1),
          "allocated GiB", round(torch.cuda.max_memory_allocated(dev_) / 2 ** 30, 1), flush=True)
    return ts


bench._timed_steps = traced
_et = torch.cuda.Event.elapsed_time
_seen = []


def _elapsed(self, other):
    v = _et(self, other)
    _seen.append(round(v, 1))
    return v


torch.cuda.Event.elapsed_time = _elapsed
r = bench.model_step(L, D, B, torch.bfloat16, dev, n_layer=n_layer, steps=steps, graphed_ok=False)
print({k: r[k] for k in ("ms_per_step", "median_ms", "min_ms")}, "event times", _seen, "retries", torch.cuda.memory_stats(dev).get("num_alloc_retries"),
      "reserved GiB", round(torch.cuda.memory_reserved(dev) / 2 ** 30, 1), "peak allocated GiB", round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1))
