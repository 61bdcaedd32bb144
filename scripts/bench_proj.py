"""in_proj + front of the shell: library GEMM + cm_pre_fwd vs the matrix-core kernel with the epilogue (csrc/proj_kernels.h).
usage: python scripts/bench_proj.py "L B D" ...   (bf16)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyena_dna_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)


def timeit(fn, n=20, w=5):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for cfg in sys.argv[1:]:
    L, B, D = (int(x) for x in cfg.split())
    g = torch.Generator(device=dev).manual_seed(0)
    u = torch.randn(B, L, D, generator=g, device=dev).to(torch.bfloat16)
    W = (torch.randn(3 * D, D, generator=g, device=dev) / D ** 0.5).to(torch.bfloat16)
    bin_, w, b = (torch.randn(3 * D, generator=g, device=dev), torch.randn(3 * D, 3, generator=g, device=dev),
                  torch.randn(3 * D, generator=g, device=dev))
    u2 = u.reshape(B * L, D)
    t_gemm = timeit(lambda: torch.mm(W, u2.t()))
    xT = torch.mm(W, u2.t()).view(3 * D, B, L)
    t_pre = timeit(lambda: _lib.cm_pre_fwd(xT, bin_, w, b, L))
    t_fused = timeit(lambda: _lib.inproj_pre_fwd(u, W, bin_, w, b, L))
    byt = (B * L * D * 2) * (1 + 3 + 1)
    print(f"L={L} B={B} D={D}: library GEMM {t_gemm:.1f} us + cm_pre_fwd {t_pre:.1f} us = {t_gemm + t_pre:.1f} us;  "
          f"fused MFMA kernel {t_fused:.1f} us ({byt / t_fused / 1e6:.2f} TB/s of its {byt / 1e9:.2f} GB, "
          f"{2 * B * L * D * 3 * D / t_fused / 1e6:.0f} TFLOP/s)", flush=True)

    # the MLP's kernels against the graph they replace (d_inner = 4 d_model)
    F = torch.nn.functional
    P, N = B * L, 4 * D
    x = u.reshape(P, D)
    W1 = (torch.randn(N, D, generator=g, device=dev) / D ** 0.5).to(torch.bfloat16)
    b1 = torch.randn(N, generator=g, device=dev).to(torch.bfloat16)
    W2 = (torch.randn(D, N, generator=g, device=dev) / N ** 0.5).to(torch.bfloat16)
    W2T = W2.t().contiguous()
    dy = torch.randn(P, D, generator=g, device=dev).to(torch.bfloat16)
    b1f = b1.float()
    t_fc1 = timeit(lambda: F.linear(x, W1, b1))
    a = F.linear(x, W1, b1)
    t_gelu = timeit(lambda: F.gelu(a, approximate="tanh"))
    t_f = timeit(lambda: _lib.mlp_fc1_gelu_fwd(x, W1, b1f))
    t_dh = timeit(lambda: torch.mm(dy, W2))
    dh = torch.mm(dy, W2)
    a_ = a.clone().requires_grad_(True)
    hh = F.gelu(a_, approximate="tanh")
    t_dg = timeit(lambda: torch.autograd.grad(hh, a_, dh, retain_graph=True))
    t_db = timeit(lambda: dh.sum(0, dtype=torch.float32))
    t_b = timeit(lambda: _lib.mlp_dh_dgelu_bwd(dy, W2T, a))
    print(f"   MLP fwd: library fc1 {t_fc1:.1f} + GELU {t_gelu:.1f} = {t_fc1 + t_gelu:.1f} us;  fused {t_f:.1f} us "
          f"({(P * D * 2 + 2 * P * N * 2) / t_f / 1e6:.2f} TB/s, {2 * P * D * N / t_f / 1e6:.0f} TFLOP/s)", flush=True)
    print(f"   MLP bwd: library dh {t_dh:.1f} + GELU' {t_dg:.1f} + bias sum {t_db:.1f} = {t_dh + t_dg + t_db:.1f} us;  fused {t_b:.1f} us "
          f"({(P * D * 2 + 2 * P * N * 2) / t_b / 1e6:.2f} TB/s)", flush=True)
    t_cs = timeit(lambda: _lib.colsum(dy))
    t_ts = timeit(lambda: dy.sum(0, dtype=torch.float32))
    print(f"   bias gradient (P x {D} column sums): torch {t_ts:.1f} us;  colsum kernel {t_cs:.1f} us ({P * D * 2 / t_cs / 1e6:.2f} TB/s)", flush=True)
