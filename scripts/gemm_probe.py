#!/usr/bin/env python
"""Diagnostic: how fast are the Hyena projections' GEMM shapes (M = B*L rows, K/N in {256, 768}) through the library
paths PyTorch-ROCm offers?  python scripts/gemm_probe.py [M]"""
import os
import sys
import torch


def timeit(fn, n=10):
    for _ in range(3):
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
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    dev = "cuda"
    for (K, N) in ((256, 768), (256, 256)):
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.05
        b = torch.randn(N, device=dev, dtype=torch.bfloat16)
        wt = w.t().contiguous()
        dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
        gb = (M * K + M * N) * 2 / 1e9
        res = {
            "linear(x,w,b)": timeit(lambda: torch.nn.functional.linear(x, w, b)),
            "linear(x,w)": timeit(lambda: torch.nn.functional.linear(x, w)),
            "mm(x,wt)": timeit(lambda: torch.mm(x, wt)),
            "addmm(b,x,wt)": timeit(lambda: torch.addmm(b, x, wt)),
            "dgrad mm(dy,w)": timeit(lambda: torch.mm(dy, w)),
            "wgrad mm(dy^T,x)": timeit(lambda: torch.mm(dy.t(), x)),
            "wgrad mm(x^T,dy)": timeit(lambda: torch.mm(x.t(), dy)),
            "wgrad splitK bmm S=16": timeit(lambda: torch.bmm(dy.view(16, M // 16, N).transpose(1, 2), x.view(16, M // 16, K)).sum(0)),
            "wgrad splitK bmm S=32": timeit(lambda: torch.bmm(dy.view(32, M // 32, N).transpose(1, 2), x.view(32, M // 32, K)).sum(0)),
            "wgrad splitK bmm S=64": timeit(lambda: torch.bmm(dy.view(64, M // 64, N).transpose(1, 2), x.view(64, M // 64, K)).sum(0)),
            "wgrad splitK bmm S=128": timeit(lambda: torch.bmm(dy.view(128, M // 128, N).transpose(1, 2), x.view(128, M // 128, K)).sum(0)),
            "wgrad splitK bmm(x^T,dy) S=32": timeit(lambda: torch.bmm(x.view(32, M // 32, K).transpose(1, 2), dy.view(32, M // 32, N)).sum(0)),
            "wgrad splitK fp32out S=32": timeit(lambda: torch.bmm(dy.view(32, M // 32, N).transpose(1, 2), x.view(32, M // 32, K), out_dtype=torch.float32).sum(0)),
            "bias grad sum(0)": timeit(lambda: dy.sum(0)),
            "bias grad fp32 ones@": timeit(lambda: torch.mm(torch.ones(1, M, device=dev, dtype=torch.bfloat16), dy)),
        }
        print(f"M={M} K={K} N={N}: min traffic {gb:.2f} GB -> {gb / 4.8:.3f} ms at 4.8 TB/s")
        for k, v in res.items():
            print(f"   {k:24s} {v:7.3f} ms")


if __name__ == "__main__":
    print("TUNABLEOP", os.environ.get("PYTORCH_TUNABLEOP_ENABLED"), "BLAS", torch.backends.cuda.preferred_blas_library())
    main()
