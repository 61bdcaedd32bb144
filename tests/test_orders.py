"""HyenaOperator at order 3 / 4 (configs/model/layer/hyena_dna.yaml:3 ships ``order: 3``; hyena.py:404-439): the recurrence runs order - 1 long
convolutions with a gate between consecutive ones and the filter emits d_model (order - 1) channels in '(v o)' order.

* the oracle (oracle/hyena_oracle.py::hyena_operator) against vectors minted from the reference's own HyenaOperator (oracle/make_golden_orders.py);
* this package's operator -- generic route (PyTorch glue around the HIP convolution) and the channel-major route of round 6 (mixer.HyenaMixerCMOrderNFunc:
  the order-2 shell kernels on three-group row views of x^T, no new kernel) -- against the same vectors, on the CPU emulation of the kernels.
"""
import os

import pytest
import torch

from oracle import hyena_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hyena_operator_orders.pt")
NAMES = ["o3_d8l64", "o3_d16l257", "o4_d8l100", "o3_d8l80_trunc"]


@pytest.fixture(scope="module")
def golden_orders():
    return torch.load(GOLDEN, weights_only=True)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_reference_vectors_at_higher_orders(golden_orders, name):
    c = golden_orders[name]
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and k in c["grads"] else v) for k, v in c["state_dict"].items()}
    for i in (3, 5):                                   # hyena.py:199: one freq parameter shared by the activations
        key = f"filter_fn.implicit_filter.{i}.freq"
        if key in sd:
            sd[key] = sd["filter_fn.implicit_filter.1.freq"]
    u = c["u"].clone().requires_grad_(True)
    y = O.hyena_operator(sd, u, l_max=c["l_max"], order=c["order"])
    y.backward(c["dy"])
    torch.testing.assert_close(y, c["y"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(u.grad, c["du"], rtol=1e-4, atol=1e-6)
    for n, g in c["grads"].items():
        torch.testing.assert_close(sd[n].grad, g, rtol=2e-4, atol=1e-5, msg=lambda m, n=n: f"{n}: {m}")


def _run(c, H):
    op = H.HyenaOperator(d_model=c["d_model"], l_max=c["l_max"], order=c["order"], filter_order=64, emb_dim=5, short_filter_order=3, modulate=True,
                         w=10, lr=6e-4, wd=0.0, lr_pos_emb=0.0)
    op.load_state_dict(c["state_dict"])
    u = c["u"].clone().requires_grad_(True)
    y = op(u)
    y.backward(c["dy"])
    return op, u, y


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("route", ["generic", "channel_major"])
def test_operator_matches_reference_vectors_at_higher_orders(emu_backend, golden_orders, name, route, monkeypatch):
    import hyena_dna_amd.hyena as H
    c = golden_orders[name]
    monkeypatch.setattr(H, "ORDER_N_FUSED", route == "channel_major")
    op, u, y = _run(c, H)
    assert op._route(c["u"].shape[1]) == ("order_n" if route == "channel_major" else "generic")
    torch.testing.assert_close(y, c["y"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(u.grad, c["du"], rtol=1e-3, atol=1e-5)
    assert set(n for n, p in op.named_parameters() if p.grad is not None) == set(c["grads"])
    for n, p in op.named_parameters():
        if p.grad is not None:
            torch.testing.assert_close(p.grad, c["grads"][n], rtol=2e-3, atol=1e-4, msg=lambda m, n=n: f"{n} ({route}): {m}")


def _oracle_f64(op, u, dy, L, order):
    sd = {k: (v.detach().double().requires_grad_(True) if v.is_floating_point() else v) for k, v in op.state_dict().items()}
    for i in (3, 5):
        sd[f"filter_fn.implicit_filter.{i}.freq"] = sd["filter_fn.implicit_filter.1.freq"]
    u_ref = u.detach().double().requires_grad_(True)
    y_ref = O.hyena_operator(sd, u_ref, l_max=L + 2, order=order)
    y_ref.backward(dy.double())
    return sd, u_ref, y_ref


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("B,L,D,order", [(3, 131, 8, 3), (2, 2049, 4, 3), (5, 77, 8, 5), (2, 100, 64, 3), (1, 70, 64, 4)])
def test_order_n_route_several_odd_length_sequences(emu_backend, monkeypatch, B, L, D, order):
    """B > 1 sequences of odd length (pitched channel rows; batch-major pitched rows between the convolutions; L = 2049: two tiles per row) in fp32
    against the oracle evaluated in float64: output, input gradient, every parameter gradient"""
    import hyena_dna_amd.hyena as H
    monkeypatch.setattr(H, "ORDER_N_FUSED", True)
    torch.manual_seed(4 + L)
    op = H.HyenaOperator(d_model=D, l_max=L + 2, order=order, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10)
    with torch.no_grad():
        op.filter_fn.bias.normal_(0, 0.5)
        op.in_proj.bias.normal_(0, 0.3)
    u = torch.randn(B, L, D).requires_grad_(True)
    dy = torch.randn(B, L, D)
    assert op._route(L) == "order_n"
    if D == 64:             # the fused filter kernels, one launch chain per convolution over every (order - 1)-th row of the last layer (filter_dl_split)
        ff = op.filter_fn
        layers = [ff.implicit_filter[i] for i in range(len(ff.implicit_filter))]
        assert ff._fused_filter_ok(L, layers, ff.pos_emb(L)[0], out_channels=D)
        ks = ff.filter_dl_split(L, order - 1)
        kk = ff.filter(L)[0].t().reshape(D, order - 1, L)
        for o, k_o in enumerate(ks):
            assert tuple(k_o.shape) == (D, L) and _rel(k_o, kk[:, o]) < 1e-5
    y = op(u)
    y.backward(dy)
    sd, u_ref, y_ref = _oracle_f64(op, u, dy, L, order)
    assert _rel(y, y_ref) < 2e-5 and _rel(u.grad, u_ref.grad) < 2e-5
    for n, p in op.named_parameters():
        if p.grad is not None and sd[n].grad is not None and sd[n].grad.norm() > 0:
            assert _rel(p.grad, sd[n].grad) < 2e-4, (n, _rel(p.grad, sd[n].grad))


def test_order_n_route_16bit_equals_the_generic_route(emu_backend, monkeypatch):
    """bf16 module and tensors, order 3, B = 3 odd-length sequences: the channel-major route against the generic route of the same module (same 16-bit
    filter -- whose own distance from an exact evaluation, sin(10 a) behind 8 bits, is what tests/test_gpu_filter.py prices -- same convolution
    kernels; the routes differ in where intermediate tensors are rounded to 16 bits)"""
    import hyena_dna_amd.hyena as H
    torch.manual_seed(4)
    B, L, D, order = 3, 131, 8, 3
    op = H.HyenaOperator(d_model=D, l_max=L + 2, order=order, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10).to(torch.bfloat16)
    u0 = torch.randn(B, L, D).to(torch.bfloat16)
    dy = torch.randn(B, L, D).to(torch.bfloat16)
    res = {}
    for route in ("order_n", "generic"):
        monkeypatch.setattr(H, "ORDER_N_FUSED", route == "order_n")
        assert op._route(L) == route
        op.zero_grad(set_to_none=True)
        u = u0.clone().requires_grad_(True)
        y = op(u)
        y.backward(dy)
        res[route] = dict(y=y.detach(), du=u.grad, **{n: p.grad for n, p in op.named_parameters() if p.grad is not None})
    assert res["order_n"].keys() == res["generic"].keys()
    for n in res["generic"]:
        assert _rel(res["order_n"][n], res["generic"][n]) < 4e-2, (n, _rel(res["order_n"][n], res["generic"][n]))


def test_filter_split_runs_one_launch_chain_per_convolution(emu_backend):
    """HyenaFilter.filter_dl_split beyond the launch-bound lengths: the fused filter kernels once per convolution over the rows o, o + n, ... of the last
    layer and of the decay rates -- values and every parameter gradient equal to the reference-shaped filter(L) split in '(v o)' order"""
    import hyena_dna_amd.hyena as H
    torch.manual_seed(2)
    D, n, L = 64, 2, 8200
    ff = H.HyenaFilter(D * n, emb_dim=5, order=64, seq_len=L + 2, w=10, modulate=True)
    layers = [ff.implicit_filter[i] for i in range(len(ff.implicit_filter))]
    assert ff._fused_filter_ok(L, layers, ff.pos_emb(L)[0], out_channels=D)
    ks = ff.filter_dl_split(L, n)
    assert len(ks) == n and all(tuple(k.shape) == (D, L) and k.is_contiguous() or k.stride(-1) == 1 for k in ks)
    g = [torch.randn(D, L, generator=torch.Generator().manual_seed(o)) * torch.linspace(1, 0.1, L) for o in range(n)]
    sum((k * g_).sum() for k, g_ in zip(ks, g)).backward()
    got = {name: p.grad.clone() for name, p in ff.named_parameters() if p.grad is not None}
    ff.zero_grad(set_to_none=True)
    kk = ff.filter(L)[0].t().reshape(D, n, L)                   # the reference's graph (hyena.py:229-238), '(v o)' channels
    for o in range(n):
        assert _rel(ks[o], kk[:, o]) < 1e-5
    sum((kk[:, o] * g[o]).sum() for o in range(n)).backward()
    assert got.keys() == {name for name, p in ff.named_parameters() if p.grad is not None} and len(got) >= 8
    for name, p in ff.named_parameters():
        if p.grad is not None:
            assert _rel(got[name], p.grad) < 2e-4, (name, _rel(got[name], p.grad))


def test_lm_with_order_3_mixers_both_routes_agree(emu_backend, monkeypatch):
    """HyenaDNALM built from the shipped layer config's ``order: 3`` (configs/model/layer/hyena_dna.yaml): the model's loss and every gradient are the
    same whether its mixers take the channel-major route or the op-by-op route -- and batches of several odd-length sequences are still padded to 64"""
    import hyena_dna_amd.hyena as H
    import hyena_dna_amd.lm as LM
    L, D = 71, 64
    layer = dict(l_max=130, order=3, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10)
    torch.manual_seed(0)
    m = LM.HyenaDNALM(d_model=D, n_layer=2, d_inner=4 * D, vocab_size=12, layer=layer, resid_dropout=0.0, embed_dropout=0.0,
                      pad_vocab_size_multiple=8, fused_dropout_add_ln=True, residual_in_fp32=True)
    ids = torch.randint(7, 11, (3, L))
    tgt = torch.roll(ids, -1, 1)
    assert m._aligned_length(ids) == 128
    params = [p for p in m.parameters() if p.requires_grad]
    res = {}
    for route in ("order_n", "generic"):
        monkeypatch.setattr(H, "ORDER_N_FUSED", route == "order_n")
        assert all(blk.mixer._route(128) == route for blk in m.backbone.layers)
        loss = m.loss(ids, tgt)
        res[route] = (loss.item(), torch.autograd.grad(loss, params, allow_unused=True))
    assert abs(res["order_n"][0] - res["generic"][0]) < 2e-6
    for a, b in zip(res["order_n"][1], res["generic"][1]):
        assert (a is None) == (b is None)
        if a is not None:
            assert _rel(a, b) < 2e-4


def test_order_n_route_empty_batch(emu_backend):
    """B = 0 through the channel-major route: an empty result that is still connected to every parameter (zero gradients, not None) -- as the reference's ops give"""
    import hyena_dna_amd.hyena as H
    op = H.HyenaOperator(d_model=8, l_max=66, order=3, filter_order=16, emb_dim=3, short_filter_order=3, modulate=True, w=10)
    u = torch.zeros(0, 64, 8, requires_grad=True)
    assert op._route(64) == "order_n"
    y = op(u)
    assert y.shape == (0, 64, 8)
    y.sum().backward()
    assert all(p.grad is None or torch.count_nonzero(p.grad) == 0 for p in op.parameters())
