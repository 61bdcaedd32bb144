"""out_proj forward: cm_post_fwd + library GEMM (round 3) vs the fused matrix-core kernel, generation 1 (round 4) and 2 (round 6), with and without
the zT side output and with the block's add + LayerNorm in the epilogue.  usage: [GEN=1|2] python scripts/bench_outproj.py "L B D" ..."""
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
    y = _lib.empty_rows((B, D), L, dt, dev)
    y.copy_(torch.randn(B, D, L, generator=g, device=dev).to(dt))
    xT = _lib.empty_cm(3 * D, B, L, dt, dev)
    xT.copy_(torch.randn(3 * D, B, L, generator=g, device=dev).to(dt))
    bin_ = torch.randn(3 * D, generator=g, device=dev) * 0.1
    w = torch.randn(3 * D, 3, generator=g, device=dev) * 0.5
    b = torch.randn(3 * D, generator=g, device=dev) * 0.1
    W = (torch.randn(D, D, generator=g, device=dev) / D ** 0.5).to(dt)
    bias = (torch.randn(D, generator=g, device=dev) * 0.1).to(dt)
    bf = bias.float()
    res = torch.randn(B * L, D, generator=g, device=dev)
    lw = 1.0 + 0.1 * torch.randn(D, generator=g, device=dev)
    lb = 0.1 * torch.randn(D, generator=g, device=dev)
    nb = B * L * D * 2

    def unfused():
        zT = _lib.cm_post_fwd(y, xT, bin_, w, b)
        m = _lib.cm_matrix(zT)
        return torch.addmm(bias, m.t(), W.t())

    t_post = timeit(lambda: _lib.cm_post_fwd(y, xT, bin_, w, b))
    t_unf = timeit(unfused)
    t_ln = timeit(lambda: _lib.add_norm_fwd(res.to(dt), res, lw, lb, 1e-5, dt))
    print(f"L={L} B={B} D={D}: cm_post_fwd {t_post:.1f} us; post + GEMM {t_unf:.1f} us; add_norm_fwd alone {t_ln:.1f} us", flush=True)
    o0 = unfused().view(B, L, D)
    z0 = _lib.cm_post_fwd(y, xT, bin_, w, b)
    outs = {}
    for gen in ([int(os.environ["GEN"])] if "GEN" in os.environ else [1, 2]):
        _lib.proj_kernel_generation(0, gen)
        t_f = timeit(lambda: _lib.outproj_gate_fwd(y, xT, bin_, w, b, W, bf, want_z=False))
        t_fz = timeit(lambda: _lib.outproj_gate_fwd(y, xT, bin_, w, b, W, bf, want_z=True))
        o1, z1 = _lib.outproj_gate_fwd(y, xT, bin_, w, b, W, bf, want_z=True)
        outs[gen] = o1
        print(f"  gen {gen}: fused {t_f:.1f} us ({3 * nb / t_f / 1e6:.2f} TB/s of y, x0, out); fused + zT {t_fz:.1f} us ({4 * nb / t_fz / 1e6:.2f} TB/s); "
              f"max |fused - unfused| / max|.| = {((o1.float() - o0.float()).abs().max() / o0.float().abs().max()).item():.2e}, "
              f"out != library {(o1 != o0).float().mean().item():.2e}; zT bitwise == cm_post_fwd: {bool(torch.equal(z1, z0))}", flush=True)

        def two():
            o, z = _lib.outproj_gate_fwd(y, xT, bin_, w, b, W, bf, want_z=True)
            return _lib.add_norm_fwd(o.view(B * L, D), res, lw, lb, 1e-5, dt) + (z,)

        t_two = timeit(two)
        t_one = timeit(lambda: _lib.outproj_gate_addnorm_fwd(y, xT, bin_, w, b, W, bf, True, res, lw, lb, 1e-5))
        a_, b_ = two(), _lib.outproj_gate_addnorm_fwd(y, xT, bin_, w, b, W, bf, True, res, lw, lb, 1e-5)
        same = all(torch.equal(p.reshape(-1), q.reshape(-1)) for p, q in zip(a_, b_))
        print(f"         out_proj (+ zT) then add + LayerNorm {t_two:.1f} us; one kernel {t_one:.1f} us ({(6 * nb + 2 * B * L * D * 4 * 1) / t_one / 1e6:.2f} TB/s "
              f"of y, x0, zT, residual in/out, normed); all five outputs bitwise equal: {same}", flush=True)
    if len(outs) == 2:
        print(f"  gen 1 vs gen 2: out identical {bool(torch.equal(outs[1], outs[2]))}, differing {(outs[1] != outs[2]).float().mean().item():.2e}")
