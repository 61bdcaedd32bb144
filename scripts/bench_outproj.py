"""out_proj forward: cm_post_fwd + library GEMM (round 3) vs the fused matrix-core kernel (round 4).  usage: python scripts/bench_outproj.py "L B D" ..."""
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
    dt = torch.bfloat16
    y = torch.randn(B, D, L, generator=g, device=dev).to(dt)
    xT = torch.randn(3 * D, B, L, generator=g, device=dev).to(dt)
    bin_ = torch.randn(3 * D, generator=g, device=dev) * 0.1
    w = torch.randn(3 * D, 3, generator=g, device=dev) * 0.5
    b = torch.randn(3 * D, generator=g, device=dev) * 0.1
    W = (torch.randn(D, D, generator=g, device=dev) / D ** 0.5).to(dt)
    bias = (torch.randn(D, generator=g, device=dev) * 0.1).to(dt)
    bf = bias.float()

    def unfused():
        zT = _lib.cm_post_fwd(y, xT, bin_, w, b)
        return torch.addmm(bias, zT.reshape(D, B * L).t(), W.t())

    t_post = timeit(lambda: _lib.cm_post_fwd(y, xT, bin_, w, b))
    t_unf = timeit(unfused)
    t_f = timeit(lambda: _lib.outproj_gate_fwd(y, xT, bin_, w, b, W, bf, want_z=False))
    t_fz = timeit(lambda: _lib.outproj_gate_fwd(y, xT, bin_, w, b, W, bf, want_z=True))
    nb = B * L * D * 2
    print(f"L={L} B={B} D={D}: cm_post_fwd {t_post:.1f} us; post + GEMM {t_unf:.1f} us; fused {t_f:.1f} us ({3 * nb / t_f / 1e6:.2f} TB/s of y, x0, out); "
          f"fused + zT {t_fz:.1f} us ({4 * nb / t_fz / 1e6:.2f} TB/s)", flush=True)
    o1, z1 = _lib.outproj_gate_fwd(y, xT, bin_, w, b, W, bf, want_z=True)
    o0 = unfused().view(B, L, D)
    print(f"   max |fused - unfused| / max|.| = {((o1.float() - o0.float()).abs().max() / o0.float().abs().max()).item():.2e}; "
          f"zT bitwise == cm_post_fwd: {bool(torch.equal(z1, _lib.cm_post_fwd(y, xT, bin_, w, b)))}")
    # round 5: the block's residual add + LayerNorm in the kernel's epilogue vs the kernel followed by add_norm_fwd
    res = torch.randn(B * L, D, generator=g, device=dev)
    lw = 1.0 + 0.1 * torch.randn(D, generator=g, device=dev)
    lb = 0.1 * torch.randn(D, generator=g, device=dev)

    def two():
        o, z = _lib.outproj_gate_fwd(y, xT, bin_, w, b, W, bf, want_z=True)
        return _lib.add_norm_fwd(o.view(B * L, D), res, lw, lb, 1e-5, dt) + (z,)

    t_two = timeit(two)
    t_one = timeit(lambda: _lib.outproj_gate_addnorm_fwd(y, xT, bin_, w, b, W, bf, True, res, lw, lb, 1e-5))
    a_, b_ = two(), _lib.outproj_gate_addnorm_fwd(y, xT, bin_, w, b, W, bf, True, res, lw, lb, 1e-5)
    same = all(torch.equal(p.reshape(-1), q.reshape(-1)) for p, q in zip(a_, b_))
    print(f"   out_proj (+ zT) then add + LayerNorm {t_two:.1f} us; one kernel {t_one:.1f} us ({(6 * nb + 2 * B * L * D * 4 * 1) / t_one / 1e6:.2f} TB/s "
          f"of y, x0, zT, residual in/out, normed); all five outputs bitwise equal: {same}", flush=True)
