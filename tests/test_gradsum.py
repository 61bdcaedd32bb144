"""Round 6: add_norm_bwd sums the columns of the dx0 it writes (hyena_dropout_add_norm_bwd_colsum) and hands them, through the one-slot side table of
hyena_dna_amd/_gradsum.py, to the backward of the linear layer in front of the norm, which needs exactly those sums as its bias gradient (out_proj:
hyena.py:440, fc2: simple_lm.py:207-211) -- instead of a streaming pass of its own over dx0."""
import pytest
import torch

from hyena_dna_amd import _gradsum


@pytest.mark.parametrize("rows,D,dt,p", [(37, 64, torch.bfloat16, 0.0), (1000, 128, torch.float16, 0.0), (4097, 256, torch.bfloat16, 0.2), (5, 512, torch.bfloat16, 0.0)])
def test_add_norm_bwd_offers_the_column_sums_of_dx0(emu_backend, rows, D, dt, p):
    g = torch.Generator().manual_seed(rows + D)
    dout = torch.randn(rows, D, generator=g).to(dt)
    dres = torch.randn(rows, D, generator=g)
    res_out = torch.randn(rows, D, generator=g)
    w = 1.0 + 0.2 * torch.randn(D, generator=g)
    mean = res_out.mean(1).contiguous()
    rstd = (1.0 / torch.sqrt(res_out.var(1, unbiased=False) + 1e-5)).contiguous()
    seed = torch.tensor([1234567], dtype=torch.int64)
    _gradsum.reset()
    dx, dr, dw, db = emu_backend.add_norm_bwd(dout, dres, res_out, w, mean, rstd, dt, need_dres=True, dropout_p=p, seed=seed)
    ref = emu_backend.add_norm_bwd(dout, dres, res_out, w, mean, rstd, dt, need_dres=True, dropout_p=p, seed=seed, offer_colsum=False)
    for a, b in zip((dx, dr, dw, db), ref):
        assert torch.equal(a, b)                                  # the third plane changes nothing else
    assert _gradsum.take(ref[0]) is None                           # the call without an offer: its dx0 is nobody's offered tensor (and the slot is untouched)
    view = dx.view(rows, D)
    got = _gradsum.take(view)                                     # a view of the same memory: a hit
    want = dx.double().sum(0)
    assert got is not None and got.dtype == torch.float32 and got.shape == (D,)
    assert ((got.double() - want).abs() <= 1e-6 * dx.double().abs().sum(0) + 1e-6).all()
    assert _gradsum.take(view) is None                             # taken once
    # through _lib.colsum: the offered sums are what it returns
    dx, dr, dw, db = emu_backend.add_norm_bwd(dout, dres, res_out, w, mean, rstd, dt, need_dres=True, dropout_p=p, seed=seed)
    before = _gradsum.stats()["hits"]
    cs = emu_backend.colsum(dx)
    assert _gradsum.stats()["hits"] == before + 1 and torch.allclose(cs.double(), want, rtol=1e-5, atol=1e-4)


def test_fp32_gradients_are_not_offered(emu_backend):
    rows, D = 64, 64
    g = torch.Generator().manual_seed(0)
    res_out = torch.randn(rows, D, generator=g)
    _gradsum.reset()
    dx, _, _, _ = emu_backend.add_norm_bwd(torch.randn(rows, D, generator=g), None, res_out, torch.ones(D), res_out.mean(1).contiguous(),
                                          torch.ones(rows), torch.float32, need_dres=False)
    assert _gradsum.take(dx) is None


def test_side_table_only_answers_for_the_very_tensor():
    _gradsum.reset()
    t = torch.randn(8, 4).to(torch.bfloat16)
    sums = t.float().sum(0)
    _gradsum.offer(t, sums)
    other = t.clone()
    assert _gradsum.take(other) is None                            # equal values, other memory
    _gradsum.offer(t, sums)
    assert _gradsum.take(t[:4]) is None                            # part of it
    _gradsum.offer(t, sums)
    t.add_(1)                                                      # edited in place after the offer
    assert _gradsum.take(t) is None
    sums = t.float().sum(0)
    _gradsum.offer(t, sums)
    ptr = t.data_ptr()
    del t                                                          # the producer's tensor is gone: whatever lives at that address now is not it
    imposter = torch.empty(8, 4, dtype=torch.bfloat16)
    if imposter.data_ptr() == ptr:
        assert _gradsum.take(imposter) is None
    t2 = torch.randn(8, 4).to(torch.bfloat16)
    _gradsum.offer(t2, t2.float().sum(0))
    assert torch.equal(_gradsum.take(t2.view(8, 4)), t2.float().sum(0)) and _gradsum.take(t2) is None


def test_lm_gradients_with_and_without_the_side_table(emu_backend, monkeypatch):
    """the model's bias gradients come out the same (fp32 summation order apart) whether out_proj's / fc2's backward takes the offered sums or makes its own pass"""
    import hyena_dna_amd.lm as LM
    L, D = 128, 128
    layer = dict(l_max=L + 2, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10)
    torch.manual_seed(0)
    m = LM.HyenaDNALM(d_model=D, n_layer=2, d_inner=4 * D, vocab_size=12, layer=layer, resid_dropout=0.0, embed_dropout=0.0,
                      pad_vocab_size_multiple=8, fused_dropout_add_ln=True, residual_in_fp32=True).to(torch.bfloat16)
    ids = torch.randint(7, 11, (2, L))
    tgt = torch.roll(ids, -1, 1)
    params = [p for p in m.parameters() if p.requires_grad]
    res = {}
    for on in (True, False):
        monkeypatch.setattr(_gradsum, "ENABLED", on)
        _gradsum.reset()
        h0 = _gradsum.stats()["hits"]
        loss = m.loss(ids, tgt)
        res[on] = (loss.item(), torch.autograd.grad(loss, params, allow_unused=True), _gradsum.stats()["hits"] - h0)
    assert res[True][2] >= 3 and res[False][2] == 0, (res[True][2], res[False][2])      # out_proj + fc2 of two layers (the last block's fc2 feeds the final norm)
    assert res[True][0] == res[False][0]
    for (n, _), a, b in zip(m.named_parameters(), res[True][1], res[False][1]):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.allclose(a.float(), b.float(), rtol=2e-2, atol=1e-3), n
