"""Randomised parity stress on the GPU (longer than the -m gpu suite wants to be): many seeded (B, D, L, dtype, chunk, saved?) cases of
both plans against the CPU oracle, each GPU result computed twice and required to be bitwise identical (timing-dependent hazards show
up as non-determinism long before they show up as a wrong value).   python scripts/gpu_stress_parity.py [seconds] [seed]"""
import os
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from hyena_dna_amd import _lib  # noqa: E402
from tests.test_gpu_parity import _inputs, _oracle, _rel  # noqa: E402  (the checker: oracle/ stays test infrastructure)

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
rng = torch.Generator().manual_seed(seed)
dts = [torch.float32, torch.bfloat16, torch.float16]
# (L range, D range, B range): small rows with many channels (fills every CU), the plan boundary, mixed-radix two-level sizes
classes = [((1, 1100), (1, 300), (1, 9)), ((1000, 9000), (1, 130), (1, 5)), ((8000, 33000), (1, 40), (1, 4)),
           ((32700, 32800), (1, 20), (1, 3)), ((32769, 70000), (1, 12), (1, 3)), ((70000, 300000), (1, 4), (1, 2))]


def ri(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=rng))


def run(u, k, bias, dout, chunk, saved):
    if saved:
        out, sv = _lib.fftconv_fwd(u, k, bias, chunk=chunk, save=True)
        du, dk, db = _lib.fftconv_bwd(dout, u, k, bias, chunk=chunk, saved=sv)
    else:
        out = _lib.fftconv_fwd(u, k, bias, chunk=chunk)
        du, dk, db = _lib.fftconv_bwd(dout, u, k, bias, chunk=chunk)
    torch.cuda.synchronize()
    return out, du, dk, db


t0, n, worst = time.time(), 0, {"out": 0.0, "du": 0.0, "dk": 0.0, "db": 0.0}
while time.time() - t0 < budget:
    (l0, l1), (d0, d1), (b0, b1) = classes[n % len(classes)]
    L, D, B = ri(l0, l1), ri(d0, d1), ri(b0, b1)
    dtype = dts[ri(0, 2)]
    chunk = ri(0, D) or None
    saved = bool(ri(0, 1))
    u, k, bias, dout = _inputs(B, D, L, dtype, seed=100000 * seed + n)
    if ri(0, 3) == 0:
        bias = torch.zeros(D)
    ud, kd, bd, gd = u.to(dev), k.to(dev), bias.to(dev), dout.to(dev)
    pitched = bool(ri(0, 1))
    if pitched:        # round 5: the layout the operator hands the convolution -- rows a multiple of 64 elements apart, NaN between them
        def rows(t):
            ld = _lib.row_pitch(t.shape[-1]) + 64 * ri(0, 1)
            buf = torch.full(t.shape[:-1] + (ld,), float("nan"), dtype=t.dtype, device=dev)
            buf[..., :t.shape[-1]] = t
            return buf[..., :t.shape[-1]]
        ud, kd, gd = rows(ud), rows(kd), rows(gd)
    a = run(ud, kd, bd, gd, chunk, saved)
    b = run(ud, kd, bd, gd, chunk, saved)
    tag = dict(case=n, B=B, D=D, L=L, dtype=str(dtype), chunk=chunk, saved=saved, pitched=pitched)
    for x, y, nm in zip(a, b, ("out", "du", "dk", "db")):
        assert torch.equal(x, y), ("NON-DETERMINISTIC " + nm, tag)
    out, du, dk, db = (t.cpu() for t in a)
    assert all(bool(torch.isfinite(t).all()) for t in (out, du, dk, db)), ("read between the rows", tag)
    r_out, r_du, r_dk, r_db = _oracle(u.float(), k, bias, dout.float())
    if dtype == torch.float32:
        e_out, e_du = _rel(out, r_out), _rel(du, r_du)
        assert e_out < 3e-6 and e_du < 3e-6, (e_out, e_du, tag)
    else:
        tol = 2 ** -8 if dtype == torch.bfloat16 else 2 ** -11
        e_out = float((out.float() - r_out).abs().max() / (r_out.abs().max() + 1e-30))
        e_du = float((du.float() - r_du).abs().max() / (r_du.abs().max() + 1e-30))
        assert (out.float() - r_out).abs().max() <= tol * r_out.abs().max() + 1e-6, (e_out, tag)
        assert (du.float() - r_du).abs().max() <= tol * r_du.abs().max() + 1e-6, (e_du, tag)
    e_dk, e_db = _rel(dk, r_dk), _rel(db, r_db)
    assert e_dk < 2e-5 and e_db < 3e-5, (e_dk, e_db, tag)
    if dtype == torch.float32:
        for nm, e in (("out", e_out), ("du", e_du), ("dk", e_dk), ("db", e_db)):
            worst[nm] = max(worst[nm], e)
    n += 1
print(f"{n} cases in {time.time() - t0:.0f} s, all bitwise deterministic and within tolerance; worst fp32 rel-L2 {worst}")
