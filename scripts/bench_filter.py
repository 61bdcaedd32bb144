#!/usr/bin/env python
"""Implicit-filter generation (HyenaFilter.filter_dl) forward + backward on the GPU: fused HIP kernels vs the module's
PyTorch-op path.   python scripts/bench_filter.py L [D]"""
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hyena_dna_amd.hyena import HyenaFilter


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    L = int(sys.argv[1])
    D = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else 256
    torch.manual_seed(0)
    f = HyenaFilter(D, emb_dim=5, order=64, seq_len=L + 2, w=10, lr_pos_emb=0.0).cuda()
    dk = torch.randn(D, L, device="cuda")

    def fused():
        f.zero_grad(set_to_none=True)
        f.filter_dl(L).backward(dk)

    def fused_fwd():
        with torch.no_grad():
            f.filter_dl(L)

    def fused16():
        f.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            k = f.filter_dl(L)
        k.backward(dk)

    def fused16_fwd():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            f.filter_dl(L)

    def generic(autocast):
        def run():
            f.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                k = f.filter(L)[0].transpose(0, 1).contiguous()
            k.backward(dk)
        return run

    if "--fused16-only" in sys.argv:
        print(f"filter L={L} D={D}: fused bf16-autocast graph fwd {timeit(fused16_fwd):.3f} ms, fwd+bwd {timeit(fused16):.3f} ms")
        return
    print(f"filter L={L} D={D}: fused fp32 fwd {timeit(fused_fwd):.3f} ms, fwd+bwd {timeit(fused):.3f} ms; fused bf16-autocast graph fwd "
          f"{timeit(fused16_fwd):.3f} ms, fwd+bwd {timeit(fused16):.3f} ms; "
          f"PyTorch ops fp32 {timeit(generic(False), 3):.3f} ms, PyTorch ops bf16 autocast {timeit(generic(True), 3):.3f} ms")


if __name__ == "__main__":
    main()
