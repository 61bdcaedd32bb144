#!/usr/bin/env python
"""Mint golden vectors of the implicit filter UNDER AUTOCAST from the REAL reference (imported from /root/reference): the graph the
trainer runs (``trainer.precision`` 16 / bf16), whose 16-bit roundings ``csrc/filter16_kernels.h`` reproduces.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python oracle/make_golden_filter_autocast.py        # writes tests/golden/hyena_filter_autocast.pt

Reference code exercised: ``HyenaFilter.filter`` (src/models/sequence/hyena.py:229-238) = PositionalEmbedding (109-131) -> the sine
MLP (199-215, Sin 96-106) -> ExponentialModulation (134-155), under ``torch.autocast('cpu', dtype)``, and autograd's backward of it.
"""
import os
# the vectors are pinned bit for bit: oneDNN's bf16 GEMMs sum in another order on AMX hosts than on AVX-512 ones -- mint (and check: tests/conftest.py)
# on the AVX-512 kernels, which both kinds of host have
os.environ.setdefault("ONEDNN_MAX_CPU_ISA", "AVX512_CORE_BF16")

import torch

from make_golden import OUT, import_reference


def main():
    ref = import_reference()
    cases = {}
    for name, D, L, emb, dtype, kw in [("d64l300_bf16", 64, 300, 5, torch.bfloat16, {}),
                                       ("d128l513_fp16", 128, 513, 3, torch.float16, {}),
                                       ("d256l200_bf16_shift", 256, 200, 7, torch.bfloat16, {"shift": 0.05}),
                                       ("d64l130_bf16_nomod", 64, 130, 5, torch.bfloat16, {"modulate": False})]:
        torch.manual_seed(7 + D + L)
        f = ref.HyenaFilter(D, emb_dim=emb, order=64, seq_len=L + 2, w=10, lr_pos_emb=1e-5, **kw)
        with torch.no_grad():                       # livelier than the default initialisation: biases of size 0.3
            for m in f.implicit_filter:
                if isinstance(m, torch.nn.Linear) and m.bias is not None:
                    m.bias.normal_(0, 0.3)
        dk = torch.randn(D, L, generator=torch.Generator().manual_seed(3))
        with torch.autocast("cpu", dtype=dtype):
            k = f.filter(L)                                            # (1, L, D)
        k.float().backward(dk.t()[None])
        cases[name] = dict(state_dict={n: v.detach().clone() for n, v in f.state_dict().items()}, D=D, L=L, emb_dim=emb, dtype=dtype,
                           kwargs=kw, dk=dk, k=k.detach()[0].t().contiguous().float(), k_dtype=k.dtype,
                           grads={n: p.grad.detach().clone() for n, p in f.named_parameters() if p.grad is not None})
    path = os.path.join(OUT, "hyena_filter_autocast.pt")
    torch.save(cases, path)
    print(path, os.path.getsize(path), {n: (c["k_dtype"], sorted(c["grads"])) for n, c in cases.items()})


if __name__ == "__main__":
    main()
