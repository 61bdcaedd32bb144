"""The matrix-core input projection with the front of the shell in its epilogue (csrc/proj_kernels.h, include/hyena_proj.h) under
tests/hipemu: xT against the fp32 product of the same 16-bit operands, vg BIT-IDENTICAL to cm_pre_fwd applied to that xT (the
kernel it replaces), ragged position counts (tiles crossing sequence boundaries, B Lx not a multiple of 64), truncation
(Lc < Lx), both widths and both 16-bit types."""
import pytest
import torch


def _case(B, Lx, D, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    u = torch.randn(B, Lx, D, generator=g).to(dtype)
    W = (torch.randn(3 * D, D, generator=g) / D ** 0.5).to(dtype)
    bin_ = torch.randn(3 * D, generator=g) * 0.3
    w = torch.randn(3 * D, 3, generator=g) * 0.5
    b = torch.randn(3 * D, generator=g) * 0.2
    return u, W, bin_, w, b


@pytest.mark.parametrize("B,Lx,Lc,D,dtype", [(1, 64, 64, 128, torch.bfloat16), (2, 100, 100, 128, torch.bfloat16), (3, 77, 70, 128, torch.float16),
                                             (1, 700, 700, 256, torch.bfloat16), (2, 333, 300, 256, torch.float16), (5, 9, 9, 128, torch.bfloat16),
                                             (1, 1500, 1500, 128, torch.bfloat16)])
def test_inproj_pre_fwd_vs_gemm_and_cm_pre(emu_backend, B, Lx, Lc, D, dtype):
    _lib = emu_backend
    u, W, bin_, w, b = _case(B, Lx, D, dtype, seed=Lx + D)
    assert _lib.proj_supported(B, Lx, D, dtype)
    xT, vg = _lib.inproj_pre_fwd(u, W, bin_, w, b, Lc)
    ref = torch.mm(W.float(), u.float().reshape(B * Lx, D).t()).view(3 * D, B, Lx)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert ((xT.float() - ref).abs() <= eps * ref.abs() + 1e-6).all()                       # one rounding of the fp32 sum
    vg_ref = _lib.cm_pre_fwd(xT, bin_, w, b, Lc)
    assert torch.equal(vg, vg_ref)
    # no bias
    xT0, vg0 = _lib.inproj_pre_fwd(u, W, None, w, b, Lc)
    assert torch.equal(xT0, xT) and torch.equal(vg0, _lib.cm_pre_fwd(xT, None, w, b, Lc))


def test_proj_supported_shapes(emu_backend):
    _lib = emu_backend
    assert not _lib.proj_supported(1, 100, 64, torch.bfloat16) and not _lib.proj_supported(1, 100, 128, torch.float32)
    assert not _lib.proj_supported(1, 4, 128, torch.bfloat16) and _lib.proj_supported(8, 32768, 256, torch.float16)


def test_operator_with_and_without_the_mfma_projection(emu_backend, monkeypatch):
    """HyenaOperator (d_model 128, bf16 tensors): the matrix-core in_proj + epilogue against the library GEMM + cm_pre_fwd path --
    the same vg bits for the same xT, so the two runs differ only by the GEMMs' own rounding of xT."""
    import hyena_dna_amd.projection as P
    from hyena_dna_amd.hyena import HyenaOperator
    torch.manual_seed(3)
    B, L, D = 2, 150, 128
    op = HyenaOperator(d_model=D, l_max=L + 2, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10,
                       lr=6e-4, wd=0.0, lr_pos_emb=0.0)
    with torch.no_grad():
        op.in_proj.bias.normal_(0, 0.1)
    op = op.to(torch.bfloat16)
    for n, buf in op.named_buffers():              # the filter's own tensors stay fp32 in training (autocast); keep them so here
        pass
    u0 = torch.randn(B, L, D).to(torch.bfloat16)
    dy = torch.randn(B, L, D).to(torch.bfloat16)
    res = []
    calls = []
    real = emu_backend.inproj_pre_fwd
    monkeypatch.setattr(emu_backend, "inproj_pre_fwd", lambda *a: (calls.append(1), real(*a))[1])
    for on in (True, False):
        monkeypatch.setattr(P, "INPROJ_MFMA", on)
        op.zero_grad(set_to_none=True)
        u = u0.clone().requires_grad_(True)
        y = op(u)
        y.backward(dy)
        res.append([y.float(), u.grad.float()] + [p.grad.float() for _, p in sorted(op.named_parameters())])
    assert len(calls) == 1                          # the kernel ran in the first pass only
    for a, b in zip(*res):
        err = ((a - b).norm() / b.norm().clamp_min(1e-20)).item()
        assert err < 2e-2, err
