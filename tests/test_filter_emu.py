"""Fused implicit-filter kernels (include/hyena_filter.h) on the CPU-emulated kernels vs the oracle's restatement of
HyenaFilter.filter (hyena.py:229-238), values and every parameter gradient."""
import pytest
import torch

from oracle import hyena_oracle as O


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def _make_filter(D, L, emb_dim=5, seed=0, **kw):
    from hyena_dna_amd.hyena import HyenaFilter
    torch.manual_seed(seed)
    f = HyenaFilter(D, emb_dim=emb_dim, order=64, seq_len=L + 2, w=10, lr_pos_emb=kw.pop("lr_pos_emb", 1e-5), **kw)
    with torch.no_grad():                       # the LM re-initialises these to N(0, 0.02); use a livelier scale
        for m in f.implicit_filter:
            if isinstance(m, torch.nn.Linear) and m.bias is not None:
                m.bias.normal_(0, 0.3)
    return f


def _oracle_filter(sd, L, dtype, modulate=True, shift=0.0):
    sd = {"filter_fn." + k: v.detach().to(dtype).requires_grad_(v.dtype.is_floating_point) for k, v in sd.items()}
    k = O.hyena_filter(sd, L, modulate=modulate, shift=shift)[0].transpose(0, 1)          # (D, L)
    return k, sd


NAMES = ["pos_emb.z", "implicit_filter.0.weight", "implicit_filter.0.bias", "implicit_filter.2.weight",
         "implicit_filter.2.bias", "implicit_filter.4.weight", "implicit_filter.4.bias", "implicit_filter.6.weight",
         "implicit_filter.1.freq"]


@pytest.mark.parametrize("D,L,emb_dim", [(64, 300, 5), (128, 512, 5), (64, 1000, 3), (256, 77, 7), (128, 1, 5)])
def test_fused_filter_matches_oracle(emu_backend, D, L, emb_dim):
    f = _make_filter(D, L, emb_dim=emb_dim, seed=D + L)
    layers = [f.implicit_filter[i] for i in range(len(f.implicit_filter))]
    assert f._fused_filter_ok(L, layers, f.pos_emb.z[:, :L])
    k = f.filter_dl(L)
    assert k.shape == (D, L) and k.dtype == torch.float32
    g = torch.Generator().manual_seed(1)
    dk = torch.randn(D, L, generator=g)
    k.backward(dk)

    truth, sd64 = _oracle_filter(f.state_dict(), L, torch.float64)
    truth.backward(dk.double())
    ref32, sd32 = _oracle_filter(f.state_dict(), L, torch.float32)
    ref32.backward(dk)
    # the kernels are fp32 like the reference's fp32 path: both sit within a few fp32 roundings (amplified by
    # sin(10 x)) of the fp64 truth; require the fused path to be no worse than 4x the reference's own error + 1e-6
    assert _rel(k, truth) < 4 * _rel(ref32, truth) + 1e-6, (_rel(k, truth), _rel(ref32, truth))
    params = dict(f.named_parameters())
    for name in NAMES:
        got = params[name].grad
        want = sd64["filter_fn." + name].grad
        ref = sd32["filter_fn." + name].grad
        if name.endswith("freq"):       # ONE Sin instance in three slots (hyena.py:199): its gradient is the sum over the slots,
            want = sum(sd64[f"filter_fn.implicit_filter.{i}.freq"].grad for i in (1, 3, 5))      # which the state-dict-keyed
            ref = sum(sd32[f"filter_fn.implicit_filter.{i}.freq"].grad for i in (1, 3, 5))       # oracle holds as three leaves
        assert got is not None, name
        if name == "pos_emb.z":
            assert torch.count_nonzero(got[:, L:]) == 0
        assert got.shape == want.shape, name
        assert _rel(got, want) < 4 * _rel(ref, want) + 2e-6, (name, _rel(got, want), _rel(ref, want))


def test_fused_filter_options(emu_backend):
    """modulate off, non-zero shift, z as a buffer (HyenaDNA: lr_pos_emb = 0), no-grad call"""
    D, L = 64, 130
    for kw, okw in (({"modulate": False}, {"modulate": False}), ({"shift": 0.05}, {"shift": 0.05}),
                    ({"lr_pos_emb": 0.0}, {})):
        f = _make_filter(D, L, seed=7, **kw)
        with torch.no_grad():
            k0 = f.filter_dl(L)
        k = f.filter_dl(L)
        assert torch.equal(k0, k)
        truth, _ = _oracle_filter(f.state_dict(), L, torch.float64, **okw)
        assert _rel(k, truth) < 1e-5, (kw, _rel(k, truth))
        k.sum().backward()
        assert f.implicit_filter[6].weight.grad is not None
        if "lr_pos_emb" in kw:
            assert not isinstance(f.pos_emb.z, torch.nn.Parameter)


def test_fused_filter_equals_generic_filter(emu_backend):
    """filter_dl (fused) and filter (PyTorch ops, the reference's graph) give the same filter"""
    f = _make_filter(128, 260, seed=11)
    a = f.filter_dl(260)
    b = f.filter(260)[0].transpose(0, 1)
    assert _rel(a, b) < 2e-5


def test_unsupported_configurations_take_the_generic_path(emu_backend):
    from hyena_dna_amd.hyena import HyenaFilter
    torch.manual_seed(0)
    for kw in ({"order": 16}, {"num_inner_mlps": 1}, {"normalized": True}, {"linear_mixer": True}):
        args = dict(emb_dim=5, order=64, seq_len=66, w=10)
        args.update(kw)
        f = HyenaFilter(64, **args)
        layers = [f.implicit_filter[i] for i in range(len(f.implicit_filter))]
        assert not f._fused_filter_ok(64, layers, f.pos_emb.z[:, :64]), kw
        k = f.filter_dl(64)
        assert _rel(k, f.filter(64)[0].transpose(0, 1)) < 1e-6
    f = HyenaFilter(48, emb_dim=5, order=64, seq_len=66, w=10)          # D not in {64, 128, 256}
    layers = [f.implicit_filter[i] for i in range(len(f.implicit_filter))]
    assert not f._fused_filter_ok(64, layers, f.pos_emb.z[:, :64])
