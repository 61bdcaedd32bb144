"""The reference op's remaining options and lengths on the HIP kernels (kernels under tests/hipemu): GELU, dropout mask, the H3 form
(v, q, head_dim), output_hbl_layout, force_fp16_output, fftfp16 -- VERDICT r3 "missing 5" -- and sequences beyond the kernels' largest
transform, served by four half-length convolutions -- "missing 6".  Reference: src/ops/fftconv.py:15-55, 58-108; oracle restatements
O.fftconv_ref / O.fftconv_h3_ref."""
import pytest
import torch

from oracle import hyena_oracle as O


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def _rows(B, H, L, seed, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    u = torch.randn(B, H, L, generator=g).to(dtype)
    k = torch.randn(H, L, generator=g) * torch.exp(-4.0 * torch.linspace(0, 1, L))[None] * 0.2
    D = torch.randn(H, generator=g)
    return u, k, D, g


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gelu_and_dropout_mask(emu_backend, dtype):
    from hyena_dna_amd.fftconv import fftconv_func
    u, k, D, g = _rows(3, 4, 700, 1, dtype)
    mask = (torch.rand(3, 4, generator=g) > 0.3).float() / 0.7
    for gelu in (True, False):
        for m in (None, mask):
            got = fftconv_func(u, k, D, dropout_mask=m, gelu=gelu)
            want = O.fftconv_ref(u, k, D, dropout_mask=m, gelu=gelu)
            assert got.dtype == dtype and got.shape == want.shape
            assert _rel(got, want) < (3e-6 if dtype == torch.float32 else 6e-3), (gelu, m is not None)


def test_gradients_through_gelu_and_mask(emu_backend):
    from hyena_dna_amd.fftconv import fftconv_func
    u, k, D, g = _rows(2, 3, 300, 2)
    mask = (torch.rand(2, 3, generator=g) > 0.5).float() * 2
    dy = torch.randn(2, 3, 300, generator=g)
    res = []
    for fn in (fftconv_func, O.fftconv_ref):
        a, b, c = (t.clone().requires_grad_(True) for t in (u, k, D))
        fn(a, b, c, dropout_mask=mask, gelu=True).backward(dy)
        res.append((a.grad, b.grad, c.grad))
    for x, y in zip(*res):
        assert _rel(x, y) < 5e-6


@pytest.mark.parametrize("head_dim", [1, 8])
@pytest.mark.parametrize("with_rev", [False, True])
def test_h3_form(emu_backend, head_dim, with_rev):
    """fftconv_func(k_in, ssm_kernel, D, v=v, head_dim=hd, q=q) == fftconv_h3_ref (src/ops/fftconv.py:37-55), values and gradients"""
    from hyena_dna_amd.fftconv import fftconv_func
    b, h, L = 2, 3, 260
    g = torch.Generator().manual_seed(5 + head_dim)
    H = h * head_dim
    kin, v, q = (torch.randn(b, H, L, generator=g) for _ in range(3))
    ssm = torch.randn(h, L, generator=g) * torch.exp(-3.0 * torch.linspace(0, 1, L))[None] * 0.3
    rev = torch.randn(h, L, generator=g) * 0.05 if with_rev else None
    D = torch.randn(h, generator=g)
    dy = torch.randn(b, H, L, generator=g)
    res = []
    for which in ("mine", "oracle"):
        t = [x.clone().requires_grad_(True) for x in (kin, ssm, D, q, v)]
        if which == "mine":
            out = fftconv_func(t[0], t[1], t[2], gelu=False, v=t[4], head_dim=head_dim, q=t[3], k_rev=rev)
        else:
            out = O.fftconv_h3_ref(t[0], t[1], t[2], t[3], t[4], head_dim=head_dim, ssm_kernel_rev=rev)
        out.backward(dy)
        res.append([out.detach()] + [x.grad for x in t])
    assert res[0][0].shape == (b, H, L)
    for x, y in zip(*res):
        assert _rel(x, y) < 1e-5


def test_output_layout_and_fp16_output_flags(emu_backend):
    from hyena_dna_amd.fftconv import fftconv_func
    u, k, D, _ = _rows(3, 4, 500, 7)
    ref = O.fftconv_ref(u, k, D, gelu=False)
    out = fftconv_func(u, k, D, gelu=False, output_hbl_layout=True)
    assert out.shape == ref.shape and out.stride() == (500, 3 * 500, 1) and _rel(out, ref) < 3e-6       # (h, b, l) in memory
    out16 = fftconv_func(u, k, D, gelu=False, force_fp16_output=True)
    assert out16.dtype == torch.float16 and _rel(out16.float(), ref) < 1e-3
    ub = u.to(torch.bfloat16)
    assert fftconv_func(ub, k, D, gelu=False, force_fp16_output=True).dtype == torch.bfloat16        # fftconv.cpp:110: bf16 input wins
    assert _rel(fftconv_func(u, k, D, gelu=False, fftfp16=True), ref) < 3e-6                           # the transform stays fp32 here
    with pytest.raises(ValueError):
        fftconv_func(u, k, D, gelu=False, q=u)                                                        # q without v: the reference kernel would ignore q
    # v without q (the reference op accepts the call): no q-multiply, i.e. the H3 form with q = 1
    got = fftconv_func(u, k, D, gelu=False, v=u)
    assert _rel(got, O.fftconv_h3_ref(u, k, D, torch.ones_like(u), u, head_dim=1)) < 1e-5


@pytest.mark.parametrize("L,dtype", [(2000, torch.float32), (2501, torch.float32), (1001, torch.float32), (3000, torch.bfloat16)])
def test_sequences_beyond_the_largest_transform(emu_backend, monkeypatch, L, dtype):
    """L > MAX_L: four half-length convolutions (recursively: MAX_L is lowered to 1000 / 700 here, L = 2501 and 3000 split twice).
    Values and all three gradients against the oracle's one long FFT; the 5-D operator layout too."""
    from hyena_dna_amd import _lib
    from hyena_dna_amd.fftconv import fftconv_func
    monkeypatch.setattr(_lib, "MAX_L", 700 if L == 2501 else 1000)
    u, k, D, g = _rows(2, 3, L, L, dtype)
    dy = torch.randn(2, 3, L, generator=g).to(dtype)
    res = []
    for fn in (lambda a, b, c: fftconv_func(a, b, c, gelu=False), lambda a, b, c: O.fftconv_ref(a, b, c, gelu=False)):
        a, b, c = (t.clone().requires_grad_(True) for t in (u, k, D))
        out = fn(a, b, c)
        out.backward(dy)
        res.append((out.detach(), a.grad, b.grad, c.grad))
    tol = 5e-6 if dtype == torch.float32 else 8e-3
    for x, y in zip(*res):
        assert x.dtype == y.dtype and _rel(x, y) < tol
    if dtype == torch.float32:
        u5 = u.reshape(2, 1, 3, 1, L)
        out5 = fftconv_func(u5, k, D.reshape(1, 3, 1), gelu=False)
        assert out5.shape == u5.shape and _rel(out5.reshape(2, 3, L), res[1][0]) < 5e-6


def test_operator_beyond_the_largest_transform_takes_the_generic_path(emu_backend, monkeypatch):
    """HyenaOperator at a length the fused core cannot take: same module, generic path, the long convolution split in four"""
    from hyena_dna_amd import _lib
    from hyena_dna_amd.hyena import HyenaOperator
    torch.manual_seed(3)
    L, Dm = 1500, 16
    op = HyenaOperator(d_model=Dm, l_max=L, order=2, filter_order=16, emb_dim=3)
    u = torch.randn(2, L, Dm)
    sd = {k_: v_.detach().clone() for k_, v_ in op.state_dict().items()}
    want = O.hyena_operator(sd, u, l_max=L)
    monkeypatch.setattr(_lib, "MAX_L", 1000)
    got = op(u)
    assert _rel(got, want) < 2e-5
