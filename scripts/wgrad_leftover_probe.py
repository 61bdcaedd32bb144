"""Round 6: the rows split_plan leaves behind its first level as ONE padded batched product (projection._leftover_product) against round 5's separate
second level + masked tail: python scripts/wgrad_leftover_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from hyena_dna_amd import _lib, projection as P  # noqa: E402

dev = torch.device("cuda", 0)


def timeit(fn, n=30, w=5):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for B, L, D in [(1, 1048575, 256), (1, 999999, 256), (8, 32767, 256), (2, 159999, 256), (256, 1023, 128), (1, 1048576, 256)]:
    dt = torch.bfloat16
    rows = B * L
    u = torch.randn(rows, D, device=dev).to(dt)
    dxT = _lib.empty_cm(3 * D, B, L, dt, dev); dxT.normal_()
    zT = _lib.empty_cm(D, B, L, dt, dev); zT.normal_()
    dy = torch.randn(rows, D, device=dev).to(dt)
    da = torch.randn(rows, 4 * D, device=dev).to(dt)
    jobs = {"dW_in": lambda: P.wgrad_cm_pm(dxT, u), "dW_out": lambda: P.wgrad_pm_cm(dy, zT), "dW1": lambda: P.split_k_weight_grad(da, u),
            "dW2": lambda: P.split_k_weight_grad(dy, da)}
    line = f"B {B} L {L} D {D} plan {P.split_plan(rows)}:"
    ref = {}
    for merged in (False, True):
        P.MERGE_LEFTOVER = merged
        line += f"\n   {'merged' if merged else 'split '}:"
        for name, fn in jobs.items():
            t = timeit(fn)
            out = fn().double()
            if not merged:
                ref[name] = out
            else:
                err = ((out - ref[name]).norm() / ref[name].norm()).item()
                assert err < 1e-5, (name, err)
            line += f"  {name} {t:7.1f} us;"
    print(line, flush=True)
    del u, dxT, zT, dy, da
    torch.cuda.empty_cache()
