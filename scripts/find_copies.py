#!/usr/bin/env python
"""Diagnostic: which lines of this package launch dtype-cast / copy kernels in one HyenaOperator layer (or one model step), with their shapes
and device time -- torch.profiler with stacks, aten::copy_ / _to_copy / clone / contiguous events grouped by the innermost hyena_dna_amd frame.
    python scripts/find_copies.py operator 1048575 1      |      python scripts/find_copies.py model 1048576 1 2
"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

what = sys.argv[1] if len(sys.argv) > 1 else "operator"
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1048575
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n_layer = int(sys.argv[4]) if len(sys.argv) > 4 else 2
D = int(sys.argv[5]) if len(sys.argv) > 5 else 256
dev = torch.device("cuda", 0)
torch.manual_seed(0)
layer = dict(l_max=L + 2, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10, lr=6e-4, wd=0.0, lr_pos_emb=0.0)
if what == "operator":
    from hyena_dna_amd.hyena import HyenaOperator
    op = HyenaOperator(d_model=D, **layer).to(dev)
    u = torch.randn(B, L, D, device=dev, dtype=torch.bfloat16, requires_grad=True)
    dy = torch.randn(B, L, D, device=dev, dtype=torch.bfloat16)

    def step():
        op.zero_grad(set_to_none=True)
        u.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = op(u)
        y.backward(dy)
else:
    from hyena_dna_amd.lm import HyenaDNALM, token_cross_entropy
    model = HyenaDNALM(d_model=D, n_layer=n_layer, d_inner=4 * D, vocab_size=12, layer=layer, resid_dropout=0.0, embed_dropout=0.1,
                       pad_vocab_size_multiple=8, fused_dropout_add_ln=True, residual_in_fp32=True).to(dev)
    opt = torch.optim.AdamW(model.parameters(), lr=6e-4, weight_decay=0.1)
    ids = torch.randint(7, 11, (B, L), device=dev)
    tgt = torch.roll(ids, -1, 1)

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = token_cross_entropy(model(ids)[0].logits, tgt)
        loss.backward()
        opt.step()

for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()

COPY_OPS = {"aten::copy_", "aten::_to_copy", "aten::clone", "aten::contiguous", "aten::to", "aten::zero_", "aten::fill_", "aten::zeros_like",
            "aten::cat", "aten::sum", "aten::add", "aten::mul", "aten::add_"}
rows = collections.defaultdict(lambda: [0, 0.0, set()])
total_kernels = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA:
        continue
    if ev.name in COPY_OPS and ev.device_time_total > 0 and not any(c.name in COPY_OPS and c.device_time_total > 0 for c in ev.cpu_children):
        frame = "?"
        for fr in ev.stack or []:
            if "hyena_dna_amd" in fr or "bench.py" in fr or "find_copies" in fr:
                frame = fr.split("/")[-1][:90]
                break
        key = (frame, ev.name)
        rows[key][0] += 1
        rows[key][1] += ev.device_time_total
        rows[key][2].add(str(ev.input_shapes)[:80])
for ev in prof.key_averages():
    if ev.device_type == torch.autograd.DeviceType.CUDA or ev.device_time_total <= 0:
        continue
print(f"== {what} L={L} B={B} D={D}: element-wise / copy ops with device time, by the innermost package frame (one step)")
for (frame, name), (n, t, shapes) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{t:9.1f} us  x{n:<3d} {name:18s} {frame:92s} {sorted(shapes)[:2]}")
print("== kernels by total device time (top 40)")
ka = [e for e in prof.key_averages() if e.device_type == torch.autograd.DeviceType.CUDA]
for e in sorted(ka, key=lambda e: -e.device_time_total)[:40]:
    print(f"{e.device_time_total:10.1f} us  x{e.count:<4d} {e.key[:120]}")
