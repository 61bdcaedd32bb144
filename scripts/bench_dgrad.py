"""out_proj's input gradient: library GEMM (dz^T) + cm_post_bwd (rounds 2 - 4) vs the fused matrix-core kernel (round 5).
usage: python scripts/bench_dgrad.py "L B D" ..."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyena_dna_amd import _lib  # noqa: E402
from hyena_dna_amd.projection import cm_from_pm  # noqa: E402

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
    dt = torch.bfloat16
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)      # noqa: E731
    dy2, Wo = rn(B * L, D).to(dt), (rn(D, D) / D ** 0.5).to(dt)
    y = _lib.empty_rows((B, D), L, dt, dev).copy_(rn(B, D, L).to(dt))
    xT = _lib.empty_cm(3 * D, B, L, dt, dev).copy_(rn(3 * D, B, L).to(dt))
    bin_, w, b = rn(3 * D) * 0.1, rn(3 * D, 3) * 0.5, rn(3 * D) * 0.1
    dxT = _lib.empty_like_cm(xT)
    part = _lib.cm_partials(xT, L)
    WoT = Wo.t().contiguous()

    def pair():
        dzT = cm_from_pm(Wo.t(), dy2, B, L)
        return _lib.cm_post_bwd(dzT, y, xT, bin_, w, b, dxT, part)

    t_gemm = timeit(lambda: cm_from_pm(Wo.t(), dy2, B, L))
    t_pair = timeit(pair)
    t_one = timeit(lambda: _lib.outproj_dgrad_gate_bwd(dy2, WoT, y, xT, bin_, w, b, dxT))
    nb = B * L * D * 2
    print(f"L={L} B={B} D={D}: library GEMM {t_gemm:.1f} us; GEMM + cm_post_bwd {t_pair:.1f} us; one kernel {t_one:.1f} us "
          f"({5 * nb / t_one / 1e6:.2f} TB/s of dy, y, x0, dyc, dxT)", flush=True)
