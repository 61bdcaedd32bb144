"""Round 6, NOT ADOPTED (profiles/r6_wgrad_plan.txt): the weight-gradient products' slice plan -- round 5's two levels (64 aligned slices + 256-row slices +
tail) against a slightly smaller slice count that leaves only a tail (63 x 16640 rows at 2^20 - 1 instead of 64 x 16320 + 15 x 256) -- at the reference
trainer's row counts.  The candidate plan is patched in here (projection.split_plan is round 5's): python scripts/wgrad_plan_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from hyena_dna_amd import _lib, projection as P  # noqa: E402

dev = torch.device("cuda", 0)
_plan5 = P.split_plan
P.PLAN_FEWER_SLICES = False


def _plan6(rows, out_elems=None):
    s = P.split_count(rows, out_elems)
    q = rows // s
    if not P.PLAN_FEWER_SLICES or (rows % s == 0 and q % 8 == 0) or q < 2 * P.TAIL_SLICE:
        return _plan5(rows, out_elems)
    for s2 in range(s, max(1, s - s // 8) - 1, -1):
        q2 = rows // s2
        q2 -= q2 % P.SLICE_ALIGN
        if q2 >= 2 * P.TAIL_SLICE and rows - s2 * q2 < P.TAIL_SLICE:
            return [(0, s2, q2)], s2 * q2
    return _plan5(rows, out_elems)


P.split_plan = _plan6


def timeit(fn, n=20, w=5):
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


for B, L, D in [(1, 1048575, 256), (1, 999999, 256), (8, 32767, 256), (256, 1023, 128), (2, 159999, 256), (1, 1048576, 256)]:
    dt = torch.bfloat16
    rows = B * L
    u = torch.randn(rows, D, device=dev).to(dt)
    dxT = _lib.empty_cm(3 * D, B, L, dt, dev); dxT.normal_()
    zT = _lib.empty_cm(D, B, L, dt, dev); zT.normal_()
    dy = torch.randn(rows, D, device=dev).to(dt)
    da = torch.randn(rows, 4 * D, device=dev).to(dt)
    jobs = {"dW_in  (cm x pm)": lambda: P.wgrad_cm_pm(dxT, u), "dW_out (pm x cm)": lambda: P.wgrad_pm_cm(dy, zT),
            "dW1    (pm x pm)": lambda: P.split_k_weight_grad(da, u), "dW2    (pm x pm)": lambda: P.split_k_weight_grad(dy, da)}
    line = f"B {B} L {L} D {D}:"
    ref = {}
    for flag in (False, True):
        P.PLAN_FEWER_SLICES = flag
        line += f"\n   fewer_slices={int(flag)} plan {P.split_plan(rows)} / {P.split_plan(rows, 65536)}:"
        for name, fn in jobs.items():
            t = timeit(fn)
            out = fn().double()
            if not flag:
                ref[name] = out
            else:
                err = ((out - ref[name]).norm() / ref[name].norm()).item()
                assert err < 1e-5, (name, err)
            line += f"  {name} {t:7.1f} us;"
    print(line, flush=True)
    del u, dxT, zT, dy, da
    torch.cuda.empty_cache()
