#!/usr/bin/env python
"""Residual add + LayerNorm forward + backward: fused HIP kernel vs the same graph in PyTorch ops.
python scripts/bench_block.py [rows] [D]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from hyena_dna_amd.block import dropout_add_layer_norm

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
D = int(sys.argv[2]) if len(sys.argv) > 2 else 256
x0 = torch.randn(1, rows, D, device="cuda").bfloat16().requires_grad_(True)
res = torch.randn(1, rows, D, device="cuda").requires_grad_(True)
ln = torch.nn.LayerNorm(D).cuda()
dout = torch.randn(1, rows, D, device="cuda").bfloat16()
dres = torch.randn(1, rows, D, device="cuda")


def fused():
    out, r = dropout_add_layer_norm(x0, res, ln.weight, ln.bias, 0.0, ln.eps, prenorm=True, residual_in_fp32=True)
    torch.autograd.backward([out, r], [dout, dres])


def unfused():
    r = x0 + res
    out = F.layer_norm(r.to(ln.weight.dtype), (D,), ln.weight, ln.bias, ln.eps).to(x0.dtype)
    torch.autograd.backward([out, r], [dout, dres])


def timeit(fn, n=10):
    for _ in range(3):
        for t in (x0, res, ln.weight, ln.bias):
            t.grad = None
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        for t in (x0, res, ln.weight, ln.bias):
            t.grad = None
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


tf, tu = timeit(fused), timeit(unfused)
gb = rows * D * (2 + 4 + 2 + 4 + 2 + 4 + 4 + 2 + 4) / 1e9      # fwd: x0, res in; out, res' out; bwd: dout, dres', res' in; dx0, dres out
print(f"add+LayerNorm rows={rows} D={D}: fused fwd+bwd {tf:.3f} ms ({gb / tf:.2f} TB/s over {gb:.2f} GB), PyTorch ops {tu:.3f} ms, x{tu / tf:.2f}")
