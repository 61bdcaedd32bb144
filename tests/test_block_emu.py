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
    torch.manual_seed(0)
    a = dropout_add_layer_norm(x0, res, w, b, 0.5, 1e-5, prenorm=True, residual_in_fp32=True)
    torch.manual_seed(0)
    dropped = F.dropout(x0, 0.5, training=True)
    assert torch.allclose(a[1], dropped + res, atol=1e-6)
    # D = 48 is outside the kernels' coverage: same graph in PyTorch ops
    y = dropout_add_layer_norm(torch.randn(2, 3, 48), torch.randn(2, 3, 48), torch.ones(48), torch.zeros(48), 0.0, 1e-5,
                               prenorm=False, residual_in_fp32=True)
    assert y.shape == (2, 3, 48)
    with pytest.raises(NotImplementedError):
        dropout_add_layer_norm(x0, res, w, b, 0.0, 1e-5, rowscale=torch.ones(2, 3))
