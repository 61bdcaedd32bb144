"""The channel-major operator shell (csrc/cm_kernels.h + projection.in_proj_cm / out_proj_cm) under tests/hipemu: values and
every gradient against the oracle's operator (a restatement of hyena.py:388-444), against the position-major path of round 1,
and through the public HyenaOperator."""
import pytest
import torch

from oracle import hyena_oracle as O


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def _ref_core_cm(xT, b_in, w, b, k, bias, L):
    """oracle pieces: xT (3D, B, Lx) -> zT (D, B, L)"""
    D = xT.shape[0] // 3
    x = (xT + b_in[:, None, None]).permute(1, 0, 2)                  # (B, 3D, Lx)
    xc = O.short_conv(x, w, b, L)
    x0, x1, v = xc.split(D, dim=1)
    return (O.fftconv_ref(v * x1, k, bias) * x0).permute(1, 0, 2)


@pytest.mark.parametrize("B,Lx,L,D,dtype", [(2, 70, 70, 8, torch.float32), (1, 2100, 2048, 6, torch.float32), (2, 130, 64, 5, torch.float32),
                                            (3, 4099, 4099, 3, torch.float32), (2, 3000, 3000, 4, torch.bfloat16),
                                            (1, 9, 9, 2, torch.float32), (2, 2049, 2049, 2, torch.float16),
                                            # several short rows per workgroup (round 6: 2 / 4 / 8 batch items of a channel share a 2048-position tile), ragged batches
                                            (5, 1023, 1023, 4, torch.bfloat16), (9, 300, 257, 3, torch.float32), (19, 130, 128, 2, torch.float16)])
def test_cm_core_vs_oracle(emu_backend, B, Lx, L, D, dtype):
    from hyena_dna_amd.mixer import hyena_mixer_core_cm
    g = torch.Generator().manual_seed(Lx + D)
    xT = torch.randn(3 * D, B, Lx, generator=g).to(dtype)
    b_in = torch.randn(3 * D, generator=g) * 0.3
    w = torch.randn(3 * D, 1, 3, generator=g) * 0.5
    b = torch.randn(3 * D, generator=g) * 0.2
    k = torch.randn(D, L, generator=g) * torch.exp(-5.0 * torch.linspace(0, 1, L))[None] * 0.1
    bias = torch.randn(D, generator=g)
    dz = torch.randn(D, B, L, generator=g).to(dtype)
    leaves = [t.clone().requires_grad_(True) for t in (xT, b_in, w, b, k, bias)]
    z = hyena_mixer_core_cm(*leaves, L)
    z.backward(dz)
    ref_leaves = [t.clone().float().requires_grad_(True) for t in (xT, b_in, w, b, k, bias)]
    zr = _ref_core_cm(*ref_leaves, L)
    zr.backward(dz.float())
    tol = 3e-6 if dtype == torch.float32 else (1.2e-2 if dtype == torch.bfloat16 else 2e-3)
    assert z.shape == (D, B, L) and z.dtype == dtype
    assert _rel(z.float(), zr) < tol
    names = ["dxT", "db_in", "dw_sc", "db_sc", "dk", "dbias"]
    for n, a, r in zip(names, leaves, ref_leaves):
        assert a.grad is not None and a.grad.shape == r.grad.shape, n
        assert _rel(a.grad.float(), r.grad) < (tol if dtype == torch.float32 else 3 * tol), (n, _rel(a.grad.float(), r.grad))
    if Lx > L:
        assert torch.count_nonzero(leaves[0].grad[:, :, L:]) == 0


def test_cm_projections_match_linear():
    """in_proj_cm / out_proj_cm = nn.Linear composed with the transposes they make unnecessary (values and gradients)"""
    from hyena_dna_amd.projection import in_proj_cm, out_proj_cm
    g = torch.Generator().manual_seed(0)
    for rows in (4096, 9999):
        u = torch.randn(2, rows, 16, generator=g).requires_grad_(True)
        W = (torch.randn(24, 16, generator=g) * 0.1).requires_grad_(True)
        bias = torch.randn(24, generator=g).requires_grad_(True)
        xT = in_proj_cm(u, W)
        d = torch.randn(xT.shape, generator=g)
        xT.backward(d)
        got = (u.grad.clone(), W.grad.clone())
        u.grad = W.grad = None
        ref = torch.nn.functional.linear(u, W).permute(2, 0, 1)
        assert _rel(xT, ref) < 1e-6
        ref.backward(d)
        assert _rel(got[0], u.grad) < 1e-6 and _rel(got[1], W.grad) < 1e-5
        zT = torch.randn(16, 2, rows, generator=g).requires_grad_(True)
        Wo = (torch.randn(24, 16, generator=g) * 0.1).requires_grad_(True)
        y = out_proj_cm(zT, Wo, bias)
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        got = (zT.grad.clone(), Wo.grad.clone(), bias.grad.clone())
        zT.grad = Wo.grad = bias.grad = None
        ref = torch.nn.functional.linear(zT.permute(1, 2, 0), Wo, bias)
        assert _rel(y, ref) < 1e-6
        ref.backward(dy)
        assert _rel(got[0], zT.grad) < 1e-6 and _rel(got[1], Wo.grad) < 1e-5 and _rel(got[2], bias.grad) < 1e-5


@pytest.mark.parametrize("name", ["d8l64", "d16l257", "d8l80_trunc"])
def test_operator_through_the_cm_path_matches_reference_goldens(emu_backend, golden_operator, name, monkeypatch):
    """HyenaOperator (D = 64 needed for the fused filter is not required here: the filter falls back to PyTorch ops at D = 8 / 16)
    in both layouts against the reference-minted operator vectors"""
    import hyena_dna_amd.hyena as H
    c = golden_operator[name]
    for cm in (True, False):
        monkeypatch.setattr(H, "CHANNEL_MAJOR", cm)
        op = H.HyenaOperator(d_model=c["d_model"], l_max=c["l_max"], order=2, filter_order=64, emb_dim=5, short_filter_order=3,
                             modulate=True, w=10, lr=6e-4, wd=0.0, lr_pos_emb=0.0)
        op.load_state_dict(c["state_dict"])
        u = c["u"].clone().requires_grad_(True)
        y = op(u)
        y.backward(c["dy"])
        torch.testing.assert_close(y, c["y"], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(u.grad, c["du"], rtol=1e-3, atol=1e-5)
        for n, p in op.named_parameters():
            torch.testing.assert_close(p.grad, c["grads"][n], rtol=2e-3, atol=1e-4, msg=lambda m, n=n: f"{n} (cm={cm}): {m}")
