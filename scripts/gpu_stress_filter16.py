"""Randomised stress of the 16-bit (autocast) filter kernels on the GPU (csrc/filter16_kernels.h): HyenaFilter.filter_dl under
torch.autocast -- forward + backward, every result computed twice and required to be bitwise identical (a timing-dependent hazard shows
up as non-determinism first) -- against the reference's own graph under the same autocast run by PyTorch's device ops (library GEMMs;
differences = 16-bit rounding flips between two fp32 summation orders, see tests/test_gpu_filter.py).
python scripts/gpu_stress_filter16.py [seconds] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from hyena_dna_amd.hyena import HyenaFilter  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
rng = torch.Generator().manual_seed(seed)


def ri(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=rng))


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def run(f, L, dk, dtype, fused):
    f.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=dtype):
        k = f.filter_dl(L) if fused else f.filter(L)[0].transpose(0, 1).float()
    k.backward(dk)
    torch.cuda.synchronize()
    return [k.detach()] + [p.grad.clone() for p in f.parameters() if p.grad is not None]


t0, n, worst_k, worst_g, worst_cols, overflowed = time.time(), 0, 0.0, 0.0, 0.0, 0
while time.time() - t0 < budget:
    D = [64, 128, 256][ri(0, 2)]
    L = [ri(1, 600), ri(600, 9000), ri(9000, 70000), ri(70000, 300000)][n % 4]
    dtype = torch.float16 if n % 5 == 4 else torch.bfloat16
    kw = dict(emb_dim=[3, 5, 7][ri(0, 2)], order=64, seq_len=L + ri(0, 3), w=[1, 10][ri(0, 1)], lr_pos_emb=[0.0, 1e-5][ri(0, 1)])
    if n % 7 == 3:
        kw["modulate"] = False
    if n % 7 == 5:
        kw["shift"] = 0.05
    torch.manual_seed(1000 * seed + n)
    f = HyenaFilter(D, **kw)
    with torch.no_grad():
        for m in f.implicit_filter:
            if isinstance(m, torch.nn.Linear) and m.bias is not None:
                m.bias.normal_(0, 0.3)
    f = f.to(dev)
    dk = torch.randn(D, L, device=dev)
    tag = dict(case=n, D=D, L=L, dtype=str(dtype), **{k_: v for k_, v in kw.items() if k_ != "order"})
    a = run(f, L, dk, dtype, True)
    b = run(f, L, dk, dtype, True)
    names = ["k"] + [nm for nm, p in f.named_parameters() if p.grad is not None]
    for nm, x, y in zip(names, a, b):
        assert torch.equal(x, y), ("NON-DETERMINISTIC", nm, tag)
    r = run(f, L, dk, dtype, False)
    assert len(r) == len(a), tag
    cols = float(((a[0] - r[0]).abs() > 1e-5 * r[0].abs().max()).any(dim=0).float().mean())
    ek = rel(a[0], r[0])
    assert cols < 0.3 and ek < 5e-2, ("k", cols, ek, tag)
    worst_k, worst_cols = max(worst_k, ek), max(worst_cols, cols)
    for nm, x, y in zip(names[1:], a[1:], r[1:]):
        assert bool(torch.isfinite(x).all()), (nm, "non-finite", tag)
        if not bool(torch.isfinite(y).all()):
            # float16: the reference's weight-gradient GEMM returns float16 and overflows on long sums (the trainer's GradScaler exists
            # for this); the kernels keep those sums in fp32
            overflowed += 1
            continue
        e = rel(x, y)
        assert e < 0.1, (nm, e, tag)
        worst_g = max(worst_g, e)
    n += 1
print(f"{n} filter cases in {time.time() - t0:.0f} s: bitwise deterministic; vs the reference graph under the same autocast (PyTorch device ops): "
      f"filter within {worst_k:.2e} rel-L2 (at most {100 * worst_cols:.1f} % of the positions touched by a rounding flip), gradients within {worst_g:.2e} ({overflowed} float16 reference gradients overflowed and were skipped)")
