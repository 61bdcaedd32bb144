import os, sys
sys.path.insert(0, "/root/repo")
import torch
from hyena_dna_amd import _lib, projection as P
dev = torch.device("cuda", 0)
def timeit(fn, n=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
plan5 = P.split_plan
def plan_even(rows, out_elems=None):
    s = P.split_count(rows, out_elems); q = rows // s
    if rows % s == 0 and q % 2 == 1 and q >= 2 * P.TAIL_SLICE:
        q -= q % P.SLICE_ALIGN
        levels, pos = [(0, s, q)], s * q
        s2 = (rows - pos) // P.TAIL_SLICE
        if s2 > 0:
            levels.append((pos, s2, P.TAIL_SLICE)); pos += s2 * P.TAIL_SLICE
        return levels, pos
    return plan5(rows, out_elems)
for B, L, D in [(1, 1000000, 256), (1, 450000, 256), (8, 4097*3, 256)]:
    dt = torch.bfloat16; rows = B * L
    u = torch.randn(rows, D, device=dev).to(dt)
    dxT = _lib.empty_cm(3 * D, B, L, dt, dev); dxT.normal_()
    zT = _lib.empty_cm(D, B, L, dt, dev); zT.normal_()
    dy = torch.randn(rows, D, device=dev).to(dt)
    da = torch.randn(rows, 4 * D, device=dev).to(dt)
    jobs = {"dW_in": lambda: P.wgrad_cm_pm(dxT, u), "dW_out": lambda: P.wgrad_pm_cm(dy, zT), "dW1": lambda: P.split_k_weight_grad(da, u), "dW2": lambda: P.split_k_weight_grad(dy, da)}
    for name, pl in (("round5", plan5), ("even", plan_even)):
        P.split_plan = pl
        print(B, L, name, pl(rows), pl(rows, 65536), {k: round(timeit(f), 1) for k, f in jobs.items()}, flush=True)
