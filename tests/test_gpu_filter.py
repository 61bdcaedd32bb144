"""MI355X parity of the fused implicit-filter kernels (include/hyena_filter.h) through the C ABI: values and every
parameter gradient against the oracle's restatement of HyenaFilter.filter (hyena.py:229-238) evaluated in fp64 on the
CPU, and -- at the full HyenaDNA lengths -- against the module's own PyTorch-op path on the same GPU; the 16-bit kernels
(torch.autocast) against the reference's graph under the same autocast."""
import pytest
import torch

from oracle import hyena_oracle as O

pytestmark = pytest.mark.gpu

NAMES = ["pos_emb.z", "implicit_filter.0.weight", "implicit_filter.0.bias", "implicit_filter.2.weight",
         "implicit_filter.2.bias", "implicit_filter.4.weight", "implicit_filter.4.bias", "implicit_filter.6.weight",
         "implicit_filter.1.freq"]


def _rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm().clamp_min(1e-30)).item()


def _make_filter(D, L, emb_dim=5, seed=0, **kw):
    from hyena_dna_amd.hyena import HyenaFilter
    torch.manual_seed(seed)
    f = HyenaFilter(D, emb_dim=emb_dim, order=64, seq_len=L + 2, w=10, lr_pos_emb=kw.pop("lr_pos_emb", 1e-5), **kw)
    with torch.no_grad():
        for m in f.implicit_filter:
            if isinstance(m, torch.nn.Linear) and m.bias is not None:
                m.bias.normal_(0, 0.3)
    return f


def _oracle_filter(sd, L, dtype):
    sd = {"filter_fn." + k: v.detach().cpu().to(dtype).requires_grad_(True) for k, v in sd.items()}
    return O.hyena_filter(sd, L)[0].transpose(0, 1), sd


@pytest.mark.parametrize("D,L,emb_dim", [(64, 300, 5), (128, 1024, 5), (256, 4099, 5), (256, 32768, 5), (128, 7, 3), (256, 1, 7)])
def test_fused_filter_vs_oracle(gpu_lib, D, L, emb_dim):
    f = _make_filter(D, L, emb_dim=emb_dim, seed=D + L).cuda()
    layers = [f.implicit_filter[i] for i in range(len(f.implicit_filter))]
    assert f._fused_filter_ok(L, layers, f.pos_emb.z[:, :L])
    k = f.filter_dl(L)
    assert k.shape == (D, L) and k.dtype == torch.float32 and k.is_cuda
    dk = torch.randn(D, L, generator=torch.Generator().manual_seed(1))
    k.backward(dk.cuda())
    truth, sd64 = _oracle_filter(f.state_dict(), L, torch.float64)
    truth.backward(dk.double())
    ref32, sd32 = _oracle_filter(f.state_dict(), L, torch.float32)
    ref32.backward(dk)
    # fp32 kernels vs the reference's fp32 path, both measured against fp64: no worse than 4x the reference's own
    # rounding error (sin(10 x) amplifies one fp32 rounding of x to ~1e-6 relative) + 1e-6
    assert _rel(k, truth) < 4 * _rel(ref32, truth) + 1e-6, (_rel(k, truth), _rel(ref32, truth))
    params = dict(f.named_parameters())
    for name in NAMES:
        got = params[name].grad
        if name.endswith("freq"):          # one Sin instance in three slots: its gradient is the sum over the slots
            want = sum(sd64[f"filter_fn.implicit_filter.{i}.freq"].grad for i in (1, 3, 5))
            ref = sum(sd32[f"filter_fn.implicit_filter.{i}.freq"].grad for i in (1, 3, 5))
        else:
            want, ref = sd64["filter_fn." + name].grad, sd32["filter_fn." + name].grad
        assert got is not None and got.shape == want.shape, name
        assert _rel(got, want) < 4 * _rel(ref, want) + 2e-6, (name, _rel(got, want), _rel(ref, want))


@pytest.mark.parametrize("D,L", [(256, 160000), (256, 450560), (256, 1048576), (256, 1048575)])
def test_fused_filter_at_hyenadna_lengths(gpu_lib, D, L):
    """full-size: fused kernels vs the module's PyTorch-op path (the reference's graph; checked against the oracle on
    the CPU by tests/test_host_logic.py) on the same device, fp32, values + gradients; and run-to-run determinism"""
    f = _make_filter(D, L, seed=3, lr_pos_emb=0.0).cuda()           # HyenaDNA: z is a buffer (lr_pos_emb = 0)
    dk = torch.randn(D, L, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
    k = f.filter_dl(L)
    k.backward(dk)
    got = {n: p.grad.clone() for n, p in f.named_parameters() if p.grad is not None}
    f.zero_grad(set_to_none=True)
    k2 = f.filter_dl(L)
    k2.backward(dk)
    assert torch.equal(k, k2)
    for n, p in f.named_parameters():
        if p.grad is not None:
            assert torch.equal(got[n], p.grad), n                   # fixed-order reductions: bitwise reproducible
    f.zero_grad(set_to_none=True)
    kr = f.filter(L)[0].transpose(0, 1)
    kr.backward(dk)
    assert _rel(k, kr) < 2e-5, _rel(k, kr)
    for n, p in f.named_parameters():
        if p.grad is not None:
            assert n in got, n
            assert _rel(got[n], p.grad) < 2e-4, (n, _rel(got[n], p.grad))      # both sum ~1e6 fp32 terms per entry


def _autocast_graph(f, L, dtype, dk):
    """the reference's own graph for the filter under autocast, evaluated by PyTorch's device ops (library GEMMs in `dtype`, fp32 sine
    and modulation by type promotion): HyenaFilter.filter (hyena.py:229-238) as the trainer runs it"""
    f.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=dtype):
        k = f.filter(L)[0].transpose(0, 1)
    assert k.dtype == torch.float32
    k.backward(dk)
    grads = {n: p.grad.clone() for n, p in f.named_parameters() if p.grad is not None}
    f.zero_grad(set_to_none=True)
    return k.detach(), grads


@pytest.mark.parametrize("D,L,emb_dim,dtype", [(64, 300, 5, torch.bfloat16), (128, 1024, 5, torch.bfloat16), (256, 4099, 5, torch.bfloat16),
                                               (256, 32768, 5, torch.bfloat16), (128, 7, 3, torch.bfloat16), (256, 1, 7, torch.bfloat16),
                                               (128, 1024, 5, torch.float16), (256, 4100, 5, torch.float16)])
def test_filter16_vs_autocast_graph(gpu_lib, D, L, emb_dim, dtype):
    """the 16-bit filter kernels (hyena_filter16_fwd / _bwd, csrc/filter16_kernels.h) under torch.autocast vs the reference's graph
    under the same autocast on the same device.  The kernels round where that graph rounds (bit-identical to the oracle under CPU
    autocast on the emulator, tests/test_filter16_emu.py), so here the two differ only where the library GEMM's and the kernel's
    different fp32 summation orders (1e-7 relative) flip a 16-bit rounding: ~1e-4 per rounded value, ~450 rounded values per position,
    and a flip moves that position's downstream by a few % (sin(10 a)).  A wrong layout or a missed rounding moves everything (the
    fp32 kernels are ~2e-1 away at these weights, asserted below)."""
    f = _make_filter(D, L, emb_dim=emb_dim, seed=D + L).cuda()
    dk = torch.randn(D, L, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    want, gref = _autocast_graph(f, L, dtype, dk)
    calls = []
    real = gpu_lib.filter_fwd
    gpu_lib.filter_fwd = lambda *a, **k_: (calls.append(k_.get("compute_dtype")), real(*a, **k_))[1]
    try:
        with torch.autocast("cuda", dtype=dtype):
            k = f.filter_dl(L)
    finally:
        gpu_lib.filter_fwd = real
    assert calls == [dtype] and k.dtype == torch.float32 and k.shape == (D, L)
    k.backward(dk)
    cols_off = ((k.detach() - want).abs() > 1e-5 * want.abs().max()).any(dim=0).float().mean().item()
    assert cols_off < 0.2, cols_off                             # positions touched by a rounding flip
    assert _rel(k, want) < 3e-2, _rel(k, want)
    if L >= 256:
        with torch.no_grad():
            k32 = f.filter_dl(L)                                    # no autocast: the fp32 kernels -- a different graph
        assert _rel(k32, want) > 3 * _rel(k, want) + 2e-3, (_rel(k32, want), _rel(k, want))
    for n, p in f.named_parameters():
        if p.grad is None:
            assert n not in gref, n
            continue
        # the reference rounds every weight / bias gradient sum to `dtype` once more (2^-9 per element in bf16); the kernels keep fp32
        e = _rel(p.grad, gref[n])
        assert e < 4e-2, (n, e)


@pytest.mark.parametrize("D,L", [(256, 160000), (256, 1048576), (256, 1048575)])
def test_filter16_at_hyenadna_lengths(gpu_lib, D, L):
    """full-size, bf16 autocast: vs the reference graph on the same device, run-to-run bitwise determinism"""
    f = _make_filter(D, L, seed=3, lr_pos_emb=0.0).cuda()
    dk = torch.randn(D, L, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
    outs = []
    for _ in range(2):
        f.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            k = f.filter_dl(L)
        k.backward(dk)
        outs.append((k.detach(), {n: p.grad.clone() for n, p in f.named_parameters() if p.grad is not None}))
    assert torch.equal(outs[0][0], outs[1][0])
    for n in outs[0][1]:
        assert torch.equal(outs[0][1][n], outs[1][1][n]), n          # fixed-order reductions: bitwise reproducible
    k, got = outs[0]
    del outs
    want, gref = _autocast_graph(f, L, torch.bfloat16, dk)
    assert ((k - want).abs() > 1e-5 * want.abs().max()).any(dim=0).float().mean().item() < 0.2
    assert _rel(k, want) < 3e-2, _rel(k, want)
    for n, g in got.items():
        # sums of ~1e6 random-sign terms: the positions whose roundings flipped and the reference's own final rounding of the sums
        assert _rel(g, gref[n]) < 6e-2, (n, _rel(g, gref[n]))


@pytest.mark.parametrize("amp_dtype", [torch.bfloat16, torch.float16])
def test_operator_uses_the_fused_filter(gpu_lib, amp_dtype, monkeypatch):
    """HyenaOperator in the HyenaDNA configuration under 16-bit autocast: the filter comes from the 16-bit kernels (the reference's
    autocast graph); with HYENA_FILTER_AUTOCAST=fp32 from the fp32 ones"""
    from hyena_dna_amd.hyena import HyenaOperator
    torch.manual_seed(0)
    op = HyenaOperator(d_model=128, l_max=1026, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True,
                       w=10, lr=6e-4, wd=0.0, lr_pos_emb=0.0).cuda()
    u = torch.randn(2, 1024, 128, device="cuda")
    k_fused = op.filter_fn.filter_dl(1024)
    k_ref = op.filter_fn.filter(1024)[0].transpose(0, 1)
    assert _rel(k_fused, k_ref) < 2e-5
    with torch.autocast("cuda", dtype=amp_dtype):
        y = op(u)
    y.float().square().mean().backward()
    assert y.dtype == amp_dtype
    grads = {n: p.grad.clone() for n, p in op.named_parameters()}
    assert all(torch.isfinite(g).all() for g in grads.values())
    # the same layer with the filter from PyTorch's ops under the same autocast (the reference graph), everything else unchanged
    op.zero_grad(set_to_none=True)
    monkeypatch.setattr(op.filter_fn, "_fused_filter_ok", lambda *a: False)
    with torch.autocast("cuda", dtype=amp_dtype):
        yg = op(u)
    yg.float().square().mean().backward()
    monkeypatch.undo()
    assert _rel(y, yg) < 3e-2, _rel(y, yg)
    for n, p in op.named_parameters():
        assert _rel(grads[n], p.grad) < 6e-2, (n, _rel(grads[n], p.grad))
    # the fp32 filter kernels under autocast (the knob): the 16-bit layer against fp32 end to end at 16-bit tolerance, as before round 3
    monkeypatch.setenv("HYENA_FILTER_AUTOCAST", "fp32")
    with torch.autocast("cuda", dtype=amp_dtype):
        ya = op(u)
    y32 = op(u)                                                  # the same layer without autocast (fp32 end to end)
    assert _rel(ya, y32) < (2e-2 if amp_dtype == torch.bfloat16 else 4e-3)


def test_split_k_projection_on_gpu(gpu_lib):
    """in_proj-shaped linear under bf16 autocast: split-K weight gradient vs an fp64 computation on the same bf16 data"""
    from hyena_dna_amd.projection import hyena_linear
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(1, 65536, 256, device="cuda", generator=g, requires_grad=True)
    lin = torch.nn.Linear(256, 768).cuda()
    dy = torch.randn(1, 65536, 768, device="cuda", generator=g)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = hyena_linear(x, lin.weight, lin.bias)
        yr = lin(x)
    assert y.dtype == torch.bfloat16 and torch.equal(y, yr)
    y.backward(dy)
    xb, dyb = x.detach().bfloat16().double()[0], dy.bfloat16().double()[0]
    assert _rel(lin.weight.grad, dyb.t() @ xb) < 4e-3            # one bf16 rounding of the fp32-accumulated result
    assert _rel(lin.bias.grad, dyb.sum(0)) < 4e-3
    assert _rel(x.grad, (dyb @ lin.weight.detach().bfloat16().double())[None]) < 4e-3


def test_split_k_projection_fp32_without_autocast(gpu_lib):
    from hyena_dna_amd.projection import hyena_linear
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(2, 32768, 256, device="cuda", generator=g, requires_grad=True)
    lin = torch.nn.Linear(256, 768).cuda()
    dy = torch.randn(2, 32768, 768, device="cuda", generator=g)
    y = hyena_linear(x, lin.weight, lin.bias)
    assert y.dtype == torch.float32 and _rel(y, lin(x)) < 1e-6
    y.backward(dy)
    got = [x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone()]
    x.grad = lin.weight.grad = lin.bias.grad = None
    lin(x).backward(dy)
    for a, b in zip(got, (x.grad, lin.weight.grad, lin.bias.grad)):
        assert _rel(a, b) < 1e-4                                  # hipBLASLt may take TF32-free fp32 paths with other orders
