"""MI355X parity of the fused residual add + LayerNorm (include/hyena_block.h) through the C ABI against the reference's
unfused graph (src/models/sequence/simple_lm.py:267-271) evaluated in fp64 on the CPU, and at a full HyenaDNA layer
size (2^20 positions x 256 channels) against the same graph in PyTorch ops on the GPU."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm().clamp_min(1e-30)).item()


def _graph(x0, residual, weight, bias, eps, out_dtype):
    res = x0.to(residual.dtype) + residual if residual is not None else x0.to(weight.dtype)
    return F.layer_norm(res.to(weight.dtype), (x0.shape[-1],), weight, bias, eps).to(out_dtype), res


@pytest.mark.parametrize("shape,dtype,with_res", [((2, 37, 64), torch.float32, True), ((3, 1000, 128), torch.bfloat16, True),
                                                  ((1, 4099, 256), torch.bfloat16, False), ((2, 513, 256), torch.float16, True),
                                                  ((1, 77, 1024), torch.bfloat16, True)])
def test_add_norm_vs_fp64_graph(gpu_lib, shape, dtype, with_res):
    from hyena_dna_amd.block import dropout_add_layer_norm
    g = torch.Generator().manual_seed(sum(shape))
    D = shape[-1]
    x0 = torch.randn(shape, generator=g).to(dtype)
    residual = torch.randn(shape, generator=g) * 2 if with_res else None
    weight, bias = 1 + 0.2 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    dout, dres = torch.randn(shape, generator=g).to(dtype), torch.randn(shape, generator=g)

    def run(dev, fused, wide):
        t = lambda a: None if a is None else (a.double() if wide else a).to(dev)      # noqa: E731
        xs = t(x0).requires_grad_(True)
        rs = None if residual is None else t(residual).requires_grad_(True)
        ws, bs = t(weight).requires_grad_(True), t(bias).requires_grad_(True)
        if fused:
            out, res = dropout_add_layer_norm(xs, rs, ws, bs, 0.0, 1e-5, prenorm=True, residual_in_fp32=True)
        else:
            out, res = _graph(xs, rs, ws, bs, 1e-5, xs.dtype)
        torch.autograd.backward([out, res], [t(dout).to(out.dtype), t(dres).to(res.dtype)])
        return [out.detach(), res.detach(), xs.grad, None if rs is None else rs.grad, ws.grad, bs.grad]

    got = run("cuda", True, False)
    want = run("cpu", False, True)
    ulp = {torch.float32: 2e-6, torch.bfloat16: 2 ** -7, torch.float16: 2 ** -10}[dtype]
    for n, a, r in zip(["out", "residual", "dx0", "dresidual", "dweight", "dbias"], got, want):
        if r is None:
            assert a is None, n
            continue
        if n in ("out", "dx0"):
            assert a.dtype == dtype
            assert (a.double().cpu() - r).abs().max() <= ulp * r.abs().max() + 1e-6, n
        else:
            assert _rel(a, r) < 5e-6, (n, _rel(a, r))


def test_add_norm_at_layer_size(gpu_lib):
    from hyena_dna_amd.block import dropout_add_layer_norm
    B, L, D = 1, 1 << 20, 256
    g = torch.Generator(device="cuda").manual_seed(0)
    x0 = torch.randn(B, L, D, device="cuda", generator=g).bfloat16().requires_grad_(True)
    residual = (torch.randn(B, L, D, device="cuda", generator=g) * 2).requires_grad_(True)
    ln = torch.nn.LayerNorm(D).cuda()
    with torch.no_grad():
        ln.weight.add_(0.2 * torch.randn(D, device="cuda", generator=g))
    dout = torch.randn(B, L, D, device="cuda", generator=g).bfloat16()
    dres = torch.randn(B, L, D, device="cuda", generator=g)

    def run(fused):
        for t in (x0, residual, ln.weight, ln.bias):
            t.grad = None
        if fused:
            out, res = dropout_add_layer_norm(x0, residual, ln.weight, ln.bias, 0.0, ln.eps, prenorm=True, residual_in_fp32=True)
        else:
            out, res = _graph(x0, residual, ln.weight, ln.bias, ln.eps, x0.dtype)
        torch.autograd.backward([out, res], [dout, dres])
        return [out.detach(), res.detach(), x0.grad.clone(), residual.grad.clone(), ln.weight.grad.clone(), ln.bias.grad.clone()]

    a, b = run(True), run(False)
    a2 = run(True)
    for n, x, y, x2 in zip(["out", "residual", "dx0", "dresidual", "dweight", "dbias"], a, b, a2):
        assert torch.equal(x, x2), n                                    # fixed-order reductions: reproducible
        if n in ("out", "dx0"):
            assert (x.float() - y.float()).abs().max() <= 2 ** -7 * y.float().abs().max() + 1e-6, n
        else:
            assert _rel(x, y) < (2e-4 if n in ("dweight", "dbias") else 5e-6), (n, _rel(x, y))   # 1e6-term fp32 sums


def test_tiny_lm_trains_on_the_gpu(gpu_lib):
    """every fused piece in one training loop under bf16 autocast on the MI355X: a 2-layer stack learns periodic DNA"""
    from tests._tiny_lm import train
    # a warmed-up, lower learning rate than the helper's default: the run is stable (round 5 ran at the helper's 3e-3, at the edge of stability, and had
    # to settle for the median of the tail -- ADVICE r5), so the LAST loss is held to the bound and so is every loss of the tail
    losses = train("cuda", steps=40, d=128, L=2048, B=4, n_layer=2, autocast_dtype=torch.bfloat16, lr=1.5e-3, warmup=8)
    assert all(l == l for l in losses)
    assert losses[-1] < 0.35 * losses[0] and max(losses[-5:]) < 0.45 * losses[0], (losses[0], losses[-5:])


def test_tiny_lm_one_step_gradients_match_the_reference_graph(gpu_lib):
    """one training step of the same stack: every parameter gradient of the fused kernels against the SAME modules evaluated by plain PyTorch ops in
    fp32 on the device (the reference's graph: F.linear / F.layer_norm / torch.fft convolution through oracle.hyena_operator) -- a gradient regression
    that a still-converging loss would hide shows here"""
    import torch.nn.functional as F
    from oracle import hyena_oracle as O
    from tests._tiny_lm import TinyLM
    torch.manual_seed(3)
    d, L, B = 128, 1024, 2
    model = TinyLM(16, d, L, 2).cuda()
    ids = torch.randint(7, 11, (B, L), device="cuda")
    tgt = torch.randint(7, 11, (B, L), device="cuda")
    logits = model(ids)
    F.cross_entropy(logits.float().reshape(-1, 16), tgt.reshape(-1)).backward()
    got = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    model.zero_grad(set_to_none=True)

    def ref_forward(m, ids):
        h, residual = m.emb(ids), None
        for blk in m.blocks:
            residual = h if residual is None else h + residual
            h = F.layer_norm(residual, (d,), blk.norm1.weight, blk.norm1.bias, blk.norm1.eps)
            h = O.hyena_operator(blk.mixer.state_dict(keep_vars=True), h, l_max=L)
            residual = h + residual
            h = F.layer_norm(residual, (d,), blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
            h = F.linear(F.gelu(F.linear(h, blk.fc1.weight, blk.fc1.bias), approximate="tanh"), blk.fc2.weight, blk.fc2.bias)
        residual = h + residual
        return F.linear(F.layer_norm(residual, (d,), m.ln_f.weight, m.ln_f.bias, m.ln_f.eps), m.emb.weight)

    F.cross_entropy(ref_forward(model, ids).float().reshape(-1, 16), tgt.reshape(-1)).backward()
    assert len(got) > 30
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        assert n in got, n
        rel = ((got[n] - p.grad).norm() / p.grad.norm().clamp_min(1e-20)).item()
        assert rel < 2e-3, (n, rel)
