#!/usr/bin/env python
"""Diagnostic (not the contract bench): one HyenaOperator layer fwd+bwd, fused HIP mixer core vs the same module
forced onto its generic PyTorch-glue path (both use the HIP long convolution)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyena_dna_amd.hyena import HyenaOperator

dev = torch.device("cuda", 0)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
D = int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4].isdigit() else 256
torch.manual_seed(0)
op = HyenaOperator(d_model=D, l_max=L + 2, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10,
                   lr=6e-4, wd=0.0, lr_pos_emb=0.0).to(dev)
u = torch.randn(B, L, D, device=dev, dtype=torch.bfloat16, requires_grad=True)
dy = torch.randn(B, L, D, device=dev)


def run(fused, n=10):
    HyenaOperator._fused_ok = (lambda self: True) if fused else (lambda self: False)
    for _ in range(3):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = op(u)
        y.backward(dy)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        op.zero_grad(set_to_none=True)       # as an optimizer step would; u is an activation in a real model
        u.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = op(u)
        y.backward(dy)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def torch_fft_conv(u, k, D, dropout_mask=None, gelu=False, force_fp16_output=False, **kw):
    """measurement aid: the reference's own long convolution (hyena.py:59-88) in torch ops, for the 'reference' mode"""
    L = u.shape[-1]
    n = 2 * L
    k_f = torch.fft.rfft(k, n=n) / n
    if u.dim() > 3:
        k_f = k_f.unsqueeze(1)                                   # (v, 1, n) against (b, h, v, z, n)
    u_f = torch.fft.rfft(u.to(k.dtype), n=n)
    y = torch.fft.irfft(u_f * k_f, n=n, norm="forward")[..., :L]
    return (y + u * D.unsqueeze(-1)).to(u.dtype)


def run_reference(n=5):
    """the whole reference graph on this GPU: PyTorch glue + torch.fft (hipFFT) long convolution"""
    import hyena_dna_amd.hyena as H
    saved = H.fftconv_func
    H.fftconv_func = torch_fft_conv
    try:
        return run(False, n)
    finally:
        H.fftconv_func = saved


mode = sys.argv[3] if len(sys.argv) > 3 else "both"
if mode == "reference":
    r = run_reference()
    print(f"HyenaOperator layer fwd+bwd  L={L} B={B} d={D} bf16 autocast: reference graph in PyTorch ops + torch.fft on this GPU {r:.3f} ms")
    sys.exit(0)
a = run(True) if mode in ("both", "fused") else float("nan")
b = run(False) if mode in ("both", "glue") else float("nan")
print(f"HyenaOperator layer fwd+bwd  L={L} B={B} d={D} bf16 autocast: fused mixer core {a:.3f} ms, PyTorch-glue path {b:.3f} ms, x{b / a:.2f}")
