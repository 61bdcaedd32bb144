"""Per-call times of the long convolution's three pieces (forward, du, dk) on the GPU, HIP events on the current stream.
usage: python scripts/oc_times.py "L B D" ...   (dtype bf16)"""
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyena_dna_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)


def timeit(fn, n=50, w=10):
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
    u = torch.randn(B, D, L, generator=g, device=dev).to(torch.bfloat16)
    k = torch.randn(D, L, generator=g, device=dev) * 0.1
    bias = torch.randn(D, generator=g, device=dev)
    dout = torch.randn(B, D, L, generator=g, device=dev).to(torch.bfloat16)
    t_f = timeit(lambda: _lib.fftconv_fwd(u, k, bias))
    t_du = timeit(lambda: _lib.fftconv_bwd(dout, u, k, bias, need_du=True, need_dk=False))
    t_dk = timeit(lambda: _lib.fftconv_bwd(dout, u, k, bias, need_du=False, need_dk=True))
    t_all = timeit(lambda: (_lib.fftconv_fwd(u, k, bias), _lib.fftconv_bwd(dout, u, k, bias)))
    def step_saved():
        _, sv = _lib.fftconv_fwd(u, k, bias, save=True)
        _lib.fftconv_bwd(dout, u, k, bias, saved=sv)
    t_sv = timeit(step_saved)
    ab = 5 * B * D * L * 2 + 12 * D * L
    print(f"L={L} B={B} D={D}: fwd {t_f:.1f} us, du {t_du:.1f} us, dk {t_dk:.1f} us, fwd+bwd {t_all:.1f} us, saved {t_sv:.1f} us, "
          f"frac {ab / (min(t_all, t_sv) * 1e-6) / 8e12:.3f}", flush=True)
