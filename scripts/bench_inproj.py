"""in_proj + front of the shell: library GEMM + cm_pre_fwd vs the matrix-core kernel, generation 1 (rounds 3 / 4) and 2 (round 6).
usage: [GEN=1|2] python scripts/bench_inproj.py "L B D" ...   (bf16)"""
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
    byt = (B * L * D * 2) * (1 + 3 + 1)
    print(f"L={L} B={B} D={D}: library GEMM {t_gemm:.1f} us + cm_pre_fwd {t_pre:.1f} us = {t_gemm + t_pre:.1f} us", flush=True)
    outs = {}
    for gen, wpg in [(1, 0), (2, 0)]:
        _lib.proj_kernel_generation(1, gen)
        t_fused = timeit(lambda: _lib.inproj_pre_fwd(u, W, bin_, w, b, L))
        x1, v1 = _lib.inproj_pre_fwd(u, W, bin_, w, b, L)
        outs[(gen, wpg)] = (x1, v1)
        same_vg = bool(torch.equal(v1, _lib.cm_pre_fwd(x1, bin_, w, b, L)))
        rel = ((x1.float() - xT.float()).norm() / xT.float().norm()).item()
        print(f"  gen {gen}{'' if not wpg else f' ({3 * wpg} wavefronts)'}: fused MFMA kernel {t_fused:.1f} us ({byt / t_fused / 1e6:.2f} TB/s of its {byt / 1e9:.2f} GB, "
              f"{2 * B * L * D * 3 * D / t_fused / 1e6:.0f} TFLOP/s); xT vs library GEMM rel {rel:.2e}, != {(x1 != xT).float().mean().item():.2e}; "
              f"vg bitwise == cm_pre_fwd(xT): {same_vg}", flush=True)
    ks = list(outs)
    print("  all variants: xT identical " + str(all(torch.equal(outs[ks[0]][0], outs[k][0]) for k in ks[1:])) +
          ", vg identical " + str(all(torch.equal(outs[ks[0]][1], outs[k][1]) for k in ks[1:])))
