#!/usr/bin/env python
"""Mint golden vectors from the REAL reference (imported from /root/reference).

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python oracle/make_golden.py            # rewrites tests/golden/*.pt

The reference has no tests or fixtures for this path, so these vectors --
outputs of the reference's own classes on seeded inputs -- are what pins the oracle
(``oracle/hyena_oracle.py``) and, through it, the HIP path.  hydra / omegaconf /
pytorch_lightning / opt_einsum are not installed here; they are replaced by inert stubs that
touch no arithmetic (recipe: SURVEY.md section 8c).
"""
import importlib
import os
import sys
import types

import torch

REF = os.environ.get("HYENA_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def import_reference():
    sys.path.insert(0, REF)

    def _stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def _get(path):
        mod, _, attr = path.rpartition(".")
        return getattr(importlib.import_module(mod), attr)

    _stub("hydra", utils=_stub("hydra.utils", get_method=_get, get_class=_get))
    _stub("omegaconf", ListConfig=list, DictConfig=type("DictConfig", (dict,), {}), OmegaConf=object)
    _stub("pytorch_lightning",
          utilities=_stub("pytorch_lightning.utilities", rank_zero_only=lambda f: f))
    _stub("opt_einsum", contract=torch.einsum)
    import src.models.sequence.hyena as ref_hyena   # noqa: E402  (reference module)
    return ref_hyena


def decaying_filter(D, L, gen):
    t = torch.linspace(0, 1, L)[None]
    return torch.randn(D, L, generator=gen) * torch.exp(-5.0 * t) * 0.1


def fftconv_cases(ref, specs=None):
    """Op-level vectors: reference fftconv_ref (hyena.py:59-88) fwd + autograd grads."""
    cases = {}
    specs = specs or [
        # name,        B, D, L,    dtype,          five_d
        ("b2d4l8",      2, 4, 8,    torch.float32,  False),
        ("b2d3l37",     2, 3, 37,   torch.float32,  False),
        ("b1d4l1023",   1, 4, 1023, torch.float32,  False),
        ("b2d4l1024",   2, 4, 1024, torch.float32,  False),
        ("b2d4l1024_5d", 2, 4, 1024, torch.float32, True),
        ("b2d4l1000_bf16", 2, 4, 1000, torch.bfloat16, True),
        ("b1d2l4100",   1, 2, 4100, torch.float32,  False),
    ]
    for name, B, D, L, dtype, five_d in specs:
        gen = torch.Generator().manual_seed(1234 + L)
        u = torch.randn(B, D, L, generator=gen).to(dtype)
        k = decaying_filter(D, L, gen)
        bias = torch.randn(D, generator=gen)
        dout = torch.randn(B, D, L, generator=gen).to(dtype)
        u_in = u.clone().requires_grad_(True)
        k_in = k.clone().requires_grad_(True)
        b_in = bias.clone().requires_grad_(True)
        if five_d:   # the shape HyenaOperator really passes (hyena.py:396-423)
            out = ref.fftconv_ref(u_in.reshape(B, 1, D, 1, L), k_in, b_in[None, :, None], None,
                                  gelu=False).reshape(B, D, L)
        else:
            out = ref.fftconv_ref(u_in, k_in, b_in, None, gelu=False)
        out.backward(dout)
        cases[name] = dict(u=u, k=k, bias=bias, dout=dout, out=out.detach(),
                           du=u_in.grad, dk=k_in.grad, dbias=b_in.grad, five_d=five_d)
    return cases


def operator_cases(ref):
    """Module-level vectors: reference HyenaOperator (hyena.py:270-448) fwd + all grads."""
    cases = {}
    for name, D, L, B, l_max in [("d8l64", 8, 64, 2, 66), ("d16l257", 16, 257, 1, 259),
                                 ("d8l80_trunc", 8, 80, 1, 64)]:
        torch.manual_seed(99 + D + L)
        op = ref.HyenaOperator(d_model=D, l_max=l_max, order=2, filter_order=64, emb_dim=5,
                               short_filter_order=3, modulate=True, w=10, lr=6e-4, wd=0.0,
                               lr_pos_emb=0.0)
        # make the filter less trivial than the default init
        u = torch.randn(B, L, D)
        u_in = u.clone().requires_grad_(True)
        y = op(u_in)
        dy = torch.randn_like(y)
        y.backward(dy)
        sd = {k_: v.detach().clone() for k_, v in op.state_dict().items()}
        grads = {n: p.grad.detach().clone() for n, p in op.named_parameters() if p.grad is not None}
        kfilt = op.filter_fn.filter(min(L, l_max)).detach()
        cases[name] = dict(state_dict=sd, u=u, y=y.detach(), dy=dy, du=u_in.grad, grads=grads,
                           k=kfilt, l_max=l_max, d_model=D)
        # bf16 autocast forward on the same weights (CPU autocast)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            y16 = op(u)
        cases[name]["y_autocast_bf16"] = y16.detach()
    return cases


def main():
    ref = import_reference()
    os.makedirs(OUT, exist_ok=True)
    torch.save(fftconv_cases(ref), os.path.join(OUT, "fftconv_ref_cases.pt"))
    # sizes beyond one 1024-point column: a two-stage column transform (L = 40000 -> M1 = 64) and a mixed-radix one
    # (L = 160000 -> M1 = 160 = 32 x 5, the hyenadna-medium-160k length), kept small by D and the 16-bit I/O
    large = [("b1d2l40000", 1, 2, 40000, torch.float32, False), ("b1d1l160000_bf16", 1, 1, 160000, torch.bfloat16, False)]
    torch.save(fftconv_cases(ref, large), os.path.join(OUT, "fftconv_ref_large.pt"))
    torch.save(operator_cases(ref), os.path.join(OUT, "hyena_operator_cases.pt"))
    meta = dict(torch=torch.__version__, reference=REF,
                note="generated by oracle/make_golden.py from the reference's own classes")
    torch.save(meta, os.path.join(OUT, "META.pt"))
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
