"""The implicit filter under 16-bit autocast (include/hyena_filter.h: hyena_filter16_fwd / _bwd) on the CPU-emulated kernels vs the
oracle's restatement of HyenaFilter.filter (hyena.py:229-238) evaluated under torch.autocast('cpu', dtype): the reference's own graph
with its own roundings.  The kernels round where that graph rounds (csrc/filter16_kernels.h), so the filter itself is compared
element-wise at fp32-accumulation noise, not at 16-bit tolerance."""
import pytest
import torch

from oracle import hyena_oracle as O
from tests.test_filter_emu import NAMES, _make_filter, _rel


def _oracle_autocast(sd, L, dtype, dk, modulate=True, shift=0.0):
    sd = {"filter_fn." + k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd.items()}
    with torch.autocast("cpu", dtype=dtype):
        k = O.hyena_filter(sd, L, modulate=modulate, shift=shift)[0].transpose(0, 1)          # (D, L)
    assert k.dtype == (torch.float32 if modulate else dtype)              # the modulation promotes (hyena.py:154)
    k = k.float()
    if dk is not None:
        k.backward(dk)
    return k.detach(), sd


@pytest.mark.parametrize("D,L,emb_dim", [(64, 300, 5), (128, 512, 5), (64, 1000, 3), (256, 77, 7), (128, 1, 5), (256, 516, 5)])
def test_filter16_bf16_is_the_autocast_graph(emu_backend, D, L, emb_dim):
    f = _make_filter(D, L, emb_dim=emb_dim, seed=D + L)
    dk = torch.randn(D, L, generator=torch.Generator().manual_seed(1))
    with torch.autocast("cpu", dtype=torch.bfloat16):
        k = f.filter_dl(L)
    assert k.shape == (D, L) and k.dtype == torch.float32
    k.backward(dk)
    want, sd = _oracle_autocast(f.state_dict(), L, torch.bfloat16, dk)
    # the filter: every rounding of the reference graph reproduced -> differences only where an fp32 accumulation-order difference
    # (~1e-7) flips a 16-bit rounding: a handful of elements at most
    d = (k.detach() - want).abs()
    assert (d > 1e-6 * want.abs().max()).float().mean().item() < 2e-3, (d > 0).float().mean().item()
    assert _rel(k.detach(), want) < 2e-3
    # fp32 is a DIFFERENT graph at these weights (sin(10 a) amplifies the 2^-9 roundings of a): the 16-bit path must not be it
    with torch.no_grad():
        k32 = f.filter_dl(L)
    if L >= 64:
        assert _rel(k32, want) > 10 * _rel(k.detach(), want) + 1e-3
    params = dict(f.named_parameters())
    for name in NAMES:
        got = params[name].grad
        ref = sd["filter_fn." + name].grad
        if name.endswith("freq"):       # ONE Sin instance in three slots (hyena.py:199)
            ref = sum(sd[f"filter_fn.implicit_filter.{i}.freq"].grad for i in (1, 3, 5))
        assert got is not None and got.shape == ref.shape, name
        # autograd rounds the weight / bias gradient sums to bf16 once more (2^-9 relative per element); the kernels keep the fp32 sums
        tol = 1e-3 if name.endswith("freq") else 6e-3
        assert _rel(got, ref) < tol, (name, _rel(got, ref))


@pytest.mark.parametrize("D,L", [(64, 300), (128, 130)])
def test_filter16_fp16(emu_backend, D, L):
    """float16 autocast (the reference trainer's `precision: 16`): the CPU's fp16 GEMMs do not accumulate like the matrix cores, so the
    comparison is at fp16 tolerance of a graph that amplifies roundings by sin(10 a), not element-exact"""
    f = _make_filter(D, L, seed=3)
    dk = torch.randn(D, L, generator=torch.Generator().manual_seed(2))
    with torch.autocast("cpu", dtype=torch.float16):
        k = f.filter_dl(L)
    k.backward(dk)
    want, sd = _oracle_autocast(f.state_dict(), L, torch.float16, dk)
    assert _rel(k.detach(), want) < 3e-3, _rel(k.detach(), want)
    params = dict(f.named_parameters())
    for name in NAMES:
        ref = sd["filter_fn." + name].grad
        if name.endswith("freq"):
            ref = sum(sd[f"filter_fn.implicit_filter.{i}.freq"].grad for i in (1, 3, 5))
        assert _rel(params[name].grad, ref) < 5e-3, (name, _rel(params[name].grad, ref))


def test_filter16_options_and_knob(emu_backend, monkeypatch):
    """modulate off, non-zero shift, z as a buffer, no-grad call; HYENA_FILTER_AUTOCAST=fp32 keeps the fp32 kernels under autocast"""
    D, L = 64, 130
    for kw, okw in (({"modulate": False}, {"modulate": False}), ({"shift": 0.05}, {"shift": 0.05}), ({"lr_pos_emb": 0.0}, {})):
        f = _make_filter(D, L, seed=7, **kw)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            with torch.no_grad():
                k0 = f.filter_dl(L)
            k = f.filter_dl(L)
        assert torch.equal(k0, k)                                    # saving the pre-activations does not change the values
        want, _ = _oracle_autocast(f.state_dict(), L, torch.bfloat16, None, **okw)
        assert _rel(k.detach(), want) < 2e-3, (kw, _rel(k.detach(), want))
        k.sum().backward()
    f = _make_filter(D, L, seed=8)
    with torch.no_grad():
        k32 = f.filter_dl(L)
        monkeypatch.setenv("HYENA_FILTER_AUTOCAST", "fp32")
        with torch.autocast("cpu", dtype=torch.bfloat16):
            ka = f.filter_dl(L)
        monkeypatch.delenv("HYENA_FILTER_AUTOCAST")
        with torch.autocast("cpu", dtype=torch.bfloat16):
            kb = f.filter_dl(L)
    assert torch.equal(ka, k32) and not torch.equal(kb, k32)


def test_filter16_is_deterministic(emu_backend):
    D, L = 128, 700
    f = _make_filter(D, L, seed=11)
    dk = torch.randn(D, L, generator=torch.Generator().manual_seed(4))
    outs = []
    for _ in range(2):
        f.zero_grad()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            k = f.filter_dl(L)
        k.backward(dk)
        outs.append([k.detach().clone()] + [p.grad.clone() for p in f.parameters() if p.grad is not None])
    assert all(torch.equal(a, b) for a, b in zip(*outs))


@pytest.mark.parametrize("name", ["d64l300_bf16", "d128l513_fp16", "d256l200_bf16_shift", "d64l130_bf16_nomod"])
def test_filter16_vs_reference_minted_autocast_vectors(emu_backend, golden_filter_autocast, name):
    """the kernels against vectors minted from the reference's own HyenaFilter under CPU autocast (oracle/make_golden_filter_autocast.py)"""
    from hyena_dna_amd.hyena import HyenaFilter
    c = golden_filter_autocast[name]
    f = HyenaFilter(c["D"], emb_dim=c["emb_dim"], order=64, seq_len=c["L"] + 2, w=10, lr_pos_emb=1e-5, **c["kwargs"])
    missing, unexpected = f.load_state_dict(c["state_dict"], strict=True)
    assert not missing and not unexpected
    with torch.autocast("cpu", dtype=c["dtype"]):
        k = f.filter_dl(c["L"])
    k.backward(c["dk"])
    bf = c["dtype"] == torch.bfloat16
    # bf16: every rounding of the reference graph reproduced (a handful of flips at most); fp16: the CPU's fp16 GEMMs accumulate
    # differently from the matrix cores, 16-bit tolerance of a graph that amplifies roundings by sin(10 a)
    assert _rel(k.detach(), c["k"]) < (2e-3 if bf else 3e-3), _rel(k.detach(), c["k"])
    if bf:
        assert ((k.detach() - c["k"]).abs() > 1e-6 * c["k"].abs().max()).float().mean().item() < 2e-3
    params = dict(f.named_parameters())
    for n, g in c["grads"].items():
        got = params[n].grad
        assert got is not None and got.shape == g.shape, n
        tol = 1e-3 if (n.endswith("freq") and bf) else 6e-3
        assert _rel(got, g) < tol, (n, _rel(got, g))
