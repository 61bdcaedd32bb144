"""Fused residual add + LayerNorm (include/hyena_block.h) on the CPU-emulated kernels vs the reference's unfused graph
(src/models/sequence/simple_lm.py:267-271): values and all gradients, prenorm and final-norm forms."""
import pytest
import torch
import torch.nn.functional as F


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def _ref(x0, residual, weight, bias, eps, prenorm):
    """simple_lm.py:267-271 with residual_in_fp32=True, evaluated in the given precision"""
    res = x0.to(residual.dtype if residual is not None else torch.float32) + residual if residual is not None else x0.float()
    out = F.layer_norm(res.to(weight.dtype), (x0.shape[-1],), weight, bias, eps)
    return (out, res) if prenorm else out


def _ref_like_pipeline(x0, residual, weight, bias, eps, prenorm, dtype):
    """... followed by the cast the next module's autocast applies to `out` (so both sides see a 16-bit `dout`)"""
    y = _ref(x0, residual, weight, bias, eps, prenorm)
    return (y[0].to(dtype), y[1]) if prenorm else y.to(dtype)


@pytest.mark.parametrize("shape,dtype,with_res,prenorm", [
    ((2, 37, 64), torch.float32, True, True), ((3, 5, 128), torch.bfloat16, True, True), ((1, 130, 256), torch.bfloat16, False, True),
    ((2, 9, 256), torch.float16, True, False), ((1, 4, 512), torch.float32, True, True), ((1, 3, 1024), torch.bfloat16, True, True),
])
def test_add_norm_matches_unfused_graph(emu_backend, shape, dtype, with_res, prenorm):
    from hyena_dna_amd.block import dropout_add_layer_norm
    g = torch.Generator().manual_seed(sum(shape))
    D = shape[-1]
    x0 = torch.randn(shape, generator=g).to(dtype)
    residual = torch.randn(shape, generator=g) * 2 if with_res else None
    weight = 1 + 0.2 * torch.randn(D, generator=g)
    bias = 0.1 * torch.randn(D, generator=g)
    dout = torch.randn(shape, generator=g)
    dres = torch.randn(shape, generator=g)

    def run(fn, cast):
        xs = x0.clone().requires_grad_(True)
        rs = None if residual is None else residual.clone().requires_grad_(True)
        ws, bs = weight.clone().requires_grad_(True), bias.clone().requires_grad_(True)
        y = fn(cast(xs), rs, ws, bs)
        if prenorm:
            out, res = y
            (out.float() * dout).sum().backward(retain_graph=True)
            (res * dres).sum().backward()
        else:
            out, res = y, None
            (out.float() * dout).sum().backward()
        return out.detach(), None if res is None else res.detach(), xs.grad, None if rs is None else rs.grad, ws.grad, bs.grad

    got = run(lambda x, r, w, b: dropout_add_layer_norm(x, r, w, b, 0.0, 1e-5, prenorm=prenorm, residual_in_fp32=True), lambda t: t)
    ref = run(lambda x, r, w, b: _ref_like_pipeline(x, r, w, b, 1e-5, prenorm, dtype), lambda t: t.float())
    names = ["out", "residual", "dx0", "dresidual", "dweight", "dbias"]
    tol16 = {torch.bfloat16: 2 ** -7, torch.float16: 2 ** -10}       # both sides are rounded to 16 bits: <= 1 ulp apart
    for n, a, r in zip(names, got, ref):
        if r is None:
            assert a is None, n
            continue
        assert a.shape == r.shape, n
        if n in ("out", "dx0") and dtype != torch.float32:
            assert a.dtype == dtype
            assert (a.float() - r.float()).abs().max() <= tol16[dtype] * r.float().abs().max() + 1e-6, n
        else:
            assert _rel(a, r) < 5e-6, (n, _rel(a, r))
    if prenorm:
        assert got[1].dtype == torch.float32


def test_dropout_and_fallbacks(emu_backend):
    from hyena_dna_amd.block import dropout_add_layer_norm
    x0 = torch.randn(2, 3, 64)
    res = torch.randn(2, 3, 64)
    w, b = torch.ones(64), torch.zeros(64)
    # D = 48 with dropout: PyTorch's dropout in front of the generic graph
    x48, r48 = torch.randn(2, 3, 48), torch.randn(2, 3, 48)
    torch.manual_seed(0)
    a = dropout_add_layer_norm(x48, r48, torch.ones(48), torch.zeros(48), 0.5, 1e-5, prenorm=True, residual_in_fp32=True)
    torch.manual_seed(0)
    dropped = F.dropout(x48, 0.5, training=True)
    assert torch.allclose(a[1], dropped + r48, atol=1e-6)
    # D = 48 is outside the kernels' coverage: same graph in PyTorch ops
    y = dropout_add_layer_norm(torch.randn(2, 3, 48), torch.randn(2, 3, 48), torch.ones(48), torch.zeros(48), 0.0, 1e-5,
                               prenorm=False, residual_in_fp32=True)
    assert y.shape == (2, 3, 48)
    with pytest.raises(NotImplementedError):
        dropout_add_layer_norm(x0, res, w, b, 0.0, 1e-5, rowscale=torch.ones(2, 3))


def _philox4x32_10(c0, c1, k0, k1):
    """Philox 4x32-10 as published (Salmon et al., SC'11), counter (c0, c1, 0, 0): the generator csrc/block_kernels.h restates"""
    M = 0xFFFFFFFF
    x = [c0, c1, 0, 0]
    for _ in range(10):
        p0, p1 = 0xD2511F53 * x[0], 0xCD9E8D57 * x[2]
        x = [(p1 >> 32) ^ x[1] ^ k0, p1 & M, (p0 >> 32) ^ x[3] ^ k1, p0 & M]
        k0, k1 = (k0 + 0x9E3779B9) & M, (k1 + 0xBB67AE85) & M
    return x


@pytest.mark.parametrize("shape,dtype,p", [((3, 70, 256), torch.float32, 0.1), ((2, 33, 128), torch.bfloat16, 0.5), ((1, 9, 64), torch.float32, 0.25),
                                           ((2, 5, 512), torch.float16, 0.1)])
def test_fused_dropout_is_a_function_of_seed_and_index(emu_backend, shape, dtype, p):
    """dropout inside the add + LayerNorm pass: the mask is exactly Philox(seed, index) >= p 2^32 (checked against a Python restatement of the
    published generator), kept elements are scaled by 1 / (1 - p), out is the LayerNorm of dropout(x0) + residual, and the backward sends
    the gradient of residual' through the SAME mask to x0 and unmasked to the residual."""
    from hyena_dna_amd import _lib
    from hyena_dna_amd.block import AddLayerNormFunc
    g = torch.Generator().manual_seed(sum(shape))
    D = shape[-1]
    x0 = (torch.randn(shape, generator=g) + 3.0).to(dtype).requires_grad_(True)              # (no zeros: a zero output means "dropped")
    residual = (torch.randn(shape, generator=g) * 2).requires_grad_(True)
    weight = (1 + 0.2 * torch.randn(D, generator=g)).requires_grad_(True)
    bias = (0.1 * torch.randn(D, generator=g)).requires_grad_(True)
    seed = torch.tensor([0x1234_5678_9ABC_DEF0 & 0x7FFF_FFFF_FFFF_FFFF], dtype=torch.int64)
    out, res = AddLayerNormFunc.apply(x0, residual, weight, bias, 1e-5, True, p, seed)
    kept = (res.detach() - residual.detach()) != 0
    # the mask, element by element, from the published generator
    sv = int(seed.item())
    k0, k1 = sv & 0xFFFFFFFF, (sv >> 32) & 0xFFFFFFFF
    thr = int(p * 4294967296.0)
    n = x0.numel()
    want = torch.empty(n, dtype=torch.bool)
    for i4 in range((n + 3) // 4):
        w = _philox4x32_10(i4 & 0xFFFFFFFF, i4 >> 32, k0, k1)
        for j in range(4):
            if 4 * i4 + j < n:
                want[4 * i4 + j] = w[j] >= thr
    assert torch.equal(kept.reshape(-1), want)
    assert abs(kept.float().mean().item() - (1 - p)) < 0.08
    scale = 1.0 / (1.0 - p)
    ref_res = x0.detach().float() * kept * scale + residual.detach()
    assert torch.allclose(res, ref_res, rtol=1e-6, atol=1e-6)
    ref_out = F.layer_norm(ref_res, (D,), weight.detach(), bias.detach(), 1e-5)
    assert _rel(out.float(), ref_out) < (1e-6 if dtype == torch.float32 else 6e-3)
    dout, dres = torch.randn(shape, generator=g).to(dtype), torch.randn(shape, generator=g)
    gx, gr, gw, gb = torch.autograd.grad([out, res], [x0, residual, weight, bias], [dout, dres])
    # reference gradients through the explicit mask
    x0r, rr = x0.detach().float().requires_grad_(True), residual.detach().requires_grad_(True)
    wr, br = weight.detach().requires_grad_(True), bias.detach().requires_grad_(True)
    res_r = x0r * kept * scale + rr
    out_r = F.layer_norm(res_r, (D,), wr, br, 1e-5)
    hx, hr, hw, hb = torch.autograd.grad([out_r, res_r], [x0r, rr, wr, br], [dout.float(), dres])
    tol = 2e-5 if dtype == torch.float32 else 8e-3
    assert _rel(gx.float(), hx) < tol and _rel(gr, hr) < 2e-5 and _rel(gw, hw) < 2e-5 and _rel(gb, hb) < 2e-5
    assert torch.equal(gx.float() != 0, kept & (hx != 0))                                    # dropped elements get exactly no gradient
    # another seed, another mask; the same seed, the same bits
    out2, res2 = AddLayerNormFunc.apply(x0, residual, weight, bias, 1e-5, True, p, seed + 1)
    assert not torch.equal((res2.detach() - residual.detach()) != 0, kept)
    out3, res3 = AddLayerNormFunc.apply(x0, residual, weight, bias, 1e-5, True, p, seed)
    assert torch.equal(out3, out) and torch.equal(res3, res)
    assert _lib.add_norm_supported(D, dtype, dtype)


def test_dropout_add_layer_norm_draws_its_seed_from_torchs_generator(emu_backend):
    from hyena_dna_amd.block import dropout_add_layer_norm
    x0, res = torch.randn(4, 16, 256) + 3.0, torch.randn(4, 16, 256)
    w, b = torch.ones(256), torch.zeros(256)
    torch.manual_seed(5)
    a = dropout_add_layer_norm(x0, res, w, b, 0.1, 1e-5, prenorm=True, residual_in_fp32=True)
    c = dropout_add_layer_norm(x0, res, w, b, 0.1, 1e-5, prenorm=True, residual_in_fp32=True)
    torch.manual_seed(5)
    a2 = dropout_add_layer_norm(x0, res, w, b, 0.1, 1e-5, prenorm=True, residual_in_fp32=True)
    assert torch.equal(a[1], a2[1]) and not torch.equal(a[1], c[1])                          # reproducible under manual_seed, fresh per call
    keep = ((a[1] - res) != 0).float().mean().item()
    assert abs(keep - 0.9) < 0.02
    with pytest.raises(ValueError):
        dropout_add_layer_norm(x0, res, w, b, 1.0, 1e-5, prenorm=True, residual_in_fp32=True)


@pytest.mark.parametrize("shape,D,V,odt,p", [((3, 70), 256, 16, torch.float32, 0.0), ((2, 33), 128, 12, torch.bfloat16, 0.25), ((1, 200), 64, 16, torch.float16, 0.1),
                                             ((4, 130), 256, 12, torch.bfloat16, 0.1)])
def test_embedding_inside_the_first_add_norm_pass(emu_backend, shape, D, V, odt, p):
    """EmbedAddLayerNormFunc == F.embedding + AddLayerNormFunc (same seed -> the same Philox mask, element for element): values bit for bit
    (the out tensor up to its one rounding), the table's gradient == the one-hot product of the unfused path's d x0, LayerNorm gradients alike."""
    from hyena_dna_amd.block import AddLayerNormFunc, EmbedAddLayerNormFunc
    g = torch.Generator().manual_seed(sum(shape) + D)
    ids = torch.randint(0, V, shape, generator=g)
    table = torch.randn(V, D, generator=g).requires_grad_(True)
    weight = (1 + 0.2 * torch.randn(D, generator=g)).requires_grad_(True)
    bias = (0.1 * torch.randn(D, generator=g)).requires_grad_(True)
    seed = torch.tensor([987654321987], dtype=torch.int64)
    args = (p, seed) if p > 0 else ()
    out, res = EmbedAddLayerNormFunc.apply(ids, table, weight, bias, 1e-5, odt, *args)
    x0 = F.embedding(ids, table)
    out_u, res_u = AddLayerNormFunc.apply(x0, None, weight, bias, 1e-5, True, *args)
    assert out.dtype == odt and res.dtype == torch.float32 and out.shape == shape + (D,)
    assert torch.equal(res, res_u)
    assert torch.equal(out, out_u.to(odt))                              # one rounding of the same fp32 LayerNorm result
    dout, dres = torch.randn(shape + (D,), generator=g).to(odt), torch.randn(shape + (D,), generator=g)
    gt, gw, gb = torch.autograd.grad([out, res], [table, weight, bias], [dout, dres])
    ht, hw, hb = torch.autograd.grad([out_u, res_u], [table, weight, bias], [dout.float(), dres])
    assert _rel(gt, ht) < 2e-6 and _rel(gw, hw) < 2e-6 and _rel(gb, hb) < 2e-6
    assert gt.shape == table.shape and (V == 16 or torch.equal(gt[V:], ht[V:]))


def test_lm_first_block_gathers_its_embedding(emu_backend, monkeypatch):
    """HyenaDNALM routes the first block through the embedding-fused pass and gets the loss and gradients of the unfused route"""
    import hyena_dna_amd.lm as LM
    L, D = 64, 64
    layer = dict(l_max=L + 2, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10)
    torch.manual_seed(0)
    m = LM.HyenaDNALM(d_model=D, n_layer=2, d_inner=4 * D, vocab_size=12, layer=layer, resid_dropout=0.0, embed_dropout=0.0,
                      pad_vocab_size_multiple=8, fused_dropout_add_ln=True, residual_in_fp32=True)
    ids = torch.randint(7, 11, (2, L))
    tgt = torch.roll(ids, -1, 1)
    calls = []
    real = LM.embedding_dropout_add_layer_norm
    monkeypatch.setattr(LM, "embedding_dropout_add_layer_norm", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    loss = m.loss(ids, tgt)
    grads = torch.autograd.grad(loss, [p for p in m.parameters() if p.requires_grad])
    assert calls == [1]
    monkeypatch.setattr(LM, "embedding_fusable", lambda *a, **k: False)
    loss_u = m.loss(ids, tgt)
    grads_u = torch.autograd.grad(loss_u, [p for p in m.parameters() if p.requires_grad])
    assert calls == [1] and abs(loss.item() - loss_u.item()) < 1e-6
    for a, b in zip(grads, grads_u):
        assert _rel(a, b) < 1e-5
