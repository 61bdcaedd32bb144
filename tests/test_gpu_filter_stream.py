"""Round 6: the implicit filter on a second stream (hyena._FilterOnSideStream) -- forward next to in_proj, backward (autograd runs a node's backward on
its forward's stream) next to the projections' gradient GEMMs.  Same kernels, same operands: results must be BIT-IDENTICAL to the one-stream run,
repeatedly (a missing wait would show as a race here), for the operator at both orders and for the language model with an optimizer step between."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run_operator(H, side, order, B, L, D, steps=3):
    H.FILTER_SIDE_STREAM = side
    torch.manual_seed(5)
    dev = torch.device("cuda", 0)
    op = H.HyenaOperator(d_model=D, l_max=L + 2, order=order, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10).to(dev)
    u = torch.randn(B, L, D, device=dev).to(torch.bfloat16)
    dy = (1e-3 * torch.randn(B, L, D, device=dev)).to(torch.bfloat16)
    opt = torch.optim.SGD(op.parameters(), lr=1e-6)          # (sums over 10^5 positions: a small step keeps the run finite)
    out = []
    for _ in range(steps):                                    # a parameter update between the steps: the second stream must see it
        opt.zero_grad(set_to_none=True)
        x = u.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = op(x)
        y.backward(dy)
        out.append([y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in op.parameters()])
        opt.step()
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("order,B,L,D", [(2, 1, 131071, 256), (2, 8, 32767, 256), (3, 2, 40000, 128), (2, 4, 8192, 128)])
def test_operator_is_bit_identical_with_the_filter_on_a_second_stream(gpu_lib, order, B, L, D):
    import hyena_dna_amd.hyena as H
    saved = H.FILTER_SIDE_STREAM
    try:
        a = _run_operator(H, True, order, B, L, D)
        b = _run_operator(H, False, order, B, L, D)
        c = _run_operator(H, True, order, B, L, D)
    finally:
        H.FILTER_SIDE_STREAM = saved
    assert H._filter_side_stream(torch.empty(1, device="cuda"), L) is not None or saved is False
    for sa, sb, sc in zip(a, b, c):
        for x, y, z in zip(sa, sb, sc):
            assert bool(torch.isfinite(x.float()).all()) and torch.equal(x, y) and torch.equal(x, z)


def test_auto_policy_of_the_second_stream(gpu_lib, monkeypatch):
    import torch.distributed as dist
    import hyena_dna_amd.hyena as H
    monkeypatch.setattr(H, "FILTER_SIDE_STREAM", "auto")
    t = torch.empty(1, device="cuda")
    assert H._filter_side_stream(t, 1 << 20) is not None and H._filter_side_stream(t, 1023) is None
    assert H._filter_side_stream(torch.empty(1), 1 << 20) is None                       # host tensors: nothing to overlap
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda *a, **k: 8)
    assert H._filter_side_stream(t, 1 << 20) is None                                    # multi-process jobs: one stream (DDP orders its buckets on it)
    monkeypatch.undo()
    monkeypatch.setattr(H, "FILTER_SIDE_STREAM", "auto")
    monkeypatch.setattr(torch.cuda, "memory_reserved", lambda *a, **k: int(0.9 * torch.cuda.get_device_properties(0).total_memory))
    assert H._filter_side_stream(t, 1 << 20) is None                                    # memory already tight: no second allocator pool
    monkeypatch.undo()
    monkeypatch.setattr(H, "FILTER_SIDE_STREAM", False)
    assert H._filter_side_stream(t, 1 << 20) is None


def test_lm_training_steps_are_bit_identical_with_the_filter_on_a_second_stream(gpu_lib):
    import hyena_dna_amd.hyena as H
    import hyena_dna_amd.lm as LM
    dev = torch.device("cuda", 0)
    L, D = 16384, 128
    layer = dict(l_max=L + 2, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10)

    def run(side):
        H.FILTER_SIDE_STREAM = side
        torch.manual_seed(0)
        m = LM.HyenaDNALM(d_model=D, n_layer=2, d_inner=4 * D, vocab_size=12, layer=layer, resid_dropout=0.0, embed_dropout=0.0,
                          pad_vocab_size_multiple=8, fused_dropout_add_ln=True, residual_in_fp32=True).to(dev)
        opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
        g = torch.Generator().manual_seed(1)
        ids = torch.randint(7, 11, (2, L), generator=g).to(dev)
        tgt = torch.roll(ids, -1, 1)
        losses = []
        for _ in range(4):
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = m.loss(ids, tgt)
            loss.backward()
            opt.step()
            losses.append(loss.item())
        torch.cuda.synchronize()
        return losses, [p.detach().clone() for p in m.parameters()]

    saved = H.FILTER_SIDE_STREAM
    try:
        la, pa = run(True)
        lb, pb = run(False)
    finally:
        H.FILTER_SIDE_STREAM = saved
    assert la == lb
    for x, y in zip(pa, pb):
        assert torch.equal(x, y)
