#!/usr/bin/env python
"""Mint golden vectors for HyenaOperator at order 3 and 4 from the REAL reference (imported from /root/reference; build container only):

    python oracle/make_golden_orders.py        # writes tests/golden/hyena_operator_orders.pt

``configs/model/layer/hyena_dna.yaml:3`` ships ``order: 3``; the recurrence of hyena.py:414-423 then runs two long convolutions with a gate between
them, the filter emits d_model (order - 1) channels in '(v o)' order (hyena.py:408-412).  The round 1-5 fixtures (make_golden.py) are order 2 only;
this script leaves them untouched.  Same stubs as make_golden.py (hydra / omegaconf / pytorch_lightning / opt_einsum: no arithmetic in them).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import OUT, import_reference  # noqa: E402


def cases(ref):
    out = {}
    for name, D, L, B, l_max, order in [("o3_d8l64", 8, 64, 2, 66, 3), ("o3_d16l257", 16, 257, 1, 259, 3), ("o4_d8l100", 8, 100, 2, 128, 4),
                                        ("o3_d8l80_trunc", 8, 80, 1, 64, 3)]:
        torch.manual_seed(7 + D + L + order)
        op = ref.HyenaOperator(d_model=D, l_max=l_max, order=order, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10,
                               lr=6e-4, wd=0.0, lr_pos_emb=0.0)
        with torch.no_grad():                         # biases away from their zero / tiny initial values, so that their gradients' index maps show
            op.filter_fn.bias.normal_(0, 0.5)
            op.in_proj.bias.normal_(0, 0.3)
            op.short_filter.bias.normal_(0, 0.3)
        u = torch.randn(B, L, D)
        u_in = u.clone().requires_grad_(True)
        y = op(u_in)
        dy = torch.randn_like(y)
        y.backward(dy)
        sd = {k_: v.detach().clone() for k_, v in op.state_dict().items()}
        grads = {n: p.grad.detach().clone() for n, p in op.named_parameters() if p.grad is not None}
        out[name] = dict(state_dict=sd, u=u, y=y.detach(), dy=dy, du=u_in.grad, grads=grads, l_max=l_max, d_model=D, order=order,
                         k=op.filter_fn.filter(min(L, l_max)).detach())
    return out


if __name__ == "__main__":
    ref = import_reference()
    path = os.path.join(OUT, "hyena_operator_orders.pt")
    torch.save(cases(ref), path)
    print(path, os.path.getsize(path))
