"""Round 6, VERDICT r5 item 5: du and dk of the workspace-free plan from ONE transform of dout (HYENA_FFTCONV_DUDK=1, dk_kernel<.., DU = true>) against the
two launches (conv_kernel with the conjugate + dk_kernel).  Times the backward alone and the forward + backward step.  usage: python scripts/bench_dudk.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyena_dna_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)


def timeit(fn, n=30, w=5):
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


for (L, B, D) in [(16384, 8, 256), (16383, 8, 256), (8192, 8, 256), (8192, 16, 256), (4096, 16, 256), (16384, 2, 256), (16384, 8, 128), (2048, 64, 128)]:
    g = torch.Generator(device=dev).manual_seed(L)
    dt = torch.bfloat16
    u = _lib.empty_rows((B, D), L, dt, dev).copy_(torch.randn(B, D, L, generator=g, device=dev).to(dt))
    dout = _lib.empty_rows((B, D), L, dt, dev).copy_(torch.randn(B, D, L, generator=g, device=dev).to(dt))
    k = _lib.empty_rows((D,), L, torch.float32, dev).copy_(torch.randn(D, L, generator=g, device=dev) * torch.exp(-5.0 * torch.linspace(0, 1, L, device=dev))[None] * 0.1)
    bias = torch.randn(D, generator=g, device=dev)
    res = {}
    line = f"L={L} B={B} D={D} bf16:"
    for knob in ("0", "1"):
        os.environ["HYENA_FFTCONV_DUDK"] = knob
        out, saved = _lib.fftconv_fwd(u, k, bias, save=True)
        t_bwd = timeit(lambda: _lib.fftconv_bwd(dout, u, k, bias, saved=saved))

        def step():
            o, s = _lib.fftconv_fwd(u, k, bias, save=True)
            return _lib.fftconv_bwd(dout, u, k, bias, saved=s)

        t_step = timeit(step)
        res[knob] = [t.clone() for t in _lib.fftconv_bwd(dout, u, k, bias, saved=saved)]
        line += f"   DUDK={knob}: backward {t_bwd:7.1f} us, fwd + bwd {t_step:7.1f} us"
    du0, dk0, db0 = res["0"]
    du1, dk1, db1 = res["1"]
    rel = ((du1.float() - du0.float()).norm() / du0.float().norm()).item()
    print(line + f"   | dk identical {bool(torch.equal(dk0, dk1))}, dbias identical {bool(torch.equal(db0, db1))}, du identical {bool(torch.equal(du0, du1))} "
                 f"(rel {rel:.1e}, differing {(du0 != du1).float().mean().item():.1e})", flush=True)
