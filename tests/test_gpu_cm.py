"""The channel-major operator shell on a real MI355X (csrc/cm_kernels.h through the C ABI): core vs the oracle pieces at sizes
with full 16-byte vectors, odd lengths (under-aligned rows), truncation (Lx > L) and every element type; and the operator in
both layouts against each other at a HyenaDNA width."""
import pytest
import torch

from oracle import hyena_oracle as O

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _ref_core_cm(xT, b_in, w, b, k, bias, L):
    D = xT.shape[0] // 3
    x = (xT + b_in[:, None, None]).permute(1, 0, 2)
    xc = O.short_conv(x, w, b, L)
    x0, x1, v = xc.split(D, dim=1)
    return (O.fftconv_ref(v * x1, k, bias) * x0).permute(1, 0, 2)


@pytest.mark.parametrize("B,Lx,L,D,dtype", [(2, 70, 70, 8, torch.float32), (2, 2100, 2048, 70, torch.float32), (1, 130, 64, 64, torch.float32),
                                            (2, 5001, 5001, 256, torch.float32), (2, 3000, 3000, 128, torch.bfloat16),
                                            (1, 40000, 40000, 64, torch.float16), (1, 160000, 160000, 16, torch.bfloat16)])
def test_cm_core_vs_oracle(gpu_lib, B, Lx, L, D, dtype):
    from hyena_dna_amd.mixer import hyena_mixer_core_cm
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(Lx + D)
    xT = torch.randn(3 * D, B, Lx, generator=g).to(dtype)
    b_in = torch.randn(3 * D, generator=g) * 0.3
    w = torch.randn(3 * D, 1, 3, generator=g) * 0.5
    b = torch.randn(3 * D, generator=g) * 0.2
    k = torch.randn(D, L, generator=g) * torch.exp(-5.0 * torch.linspace(0, 1, L))[None] * 0.1
    bias = torch.randn(D, generator=g)
    dz = torch.randn(D, B, L, generator=g).to(dtype)
    leaves = [t.to(dev).requires_grad_(True) for t in (xT, b_in, w, b, k, bias)]
    z = hyena_mixer_core_cm(*leaves, L)
    z.backward(dz.to(dev))
    ref_leaves = [t.clone().float().requires_grad_(True) for t in (xT, b_in, w, b, k, bias)]
    zr = _ref_core_cm(*ref_leaves, L)
    zr.backward(dz.float())
    tol = 3e-6 if dtype == torch.float32 else (1.2e-2 if dtype == torch.bfloat16 else 2e-3)
    assert _rel(z.float(), zr) < tol
    for n, a, r in zip(["dxT", "db_in", "dw_sc", "db_sc", "dk", "dbias"], leaves, ref_leaves):
        e = _rel(a.grad.float(), r.grad)
        assert e < (3 * tol if dtype != torch.float32 else (1e-5 if n in ("db_in", "dw_sc", "db_sc", "dbias") else tol)), (n, e)
    if Lx > L:
        assert torch.count_nonzero(leaves[0].grad[:, :, L:]) == 0


def test_operator_layouts_agree_on_gpu(gpu_lib, monkeypatch):
    """HyenaOperator at d_model = 256 (fused filter + long conv + shell), bf16 autocast: channel-major vs position-major path"""
    import hyena_dna_amd.hyena as H
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    B, L, D = 2, 6000, 256
    u0 = torch.randn(B, L, D, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(B, L, D, device=dev, dtype=torch.bfloat16)
    res = []
    sd = None
    for cm in (True, False):
        monkeypatch.setattr(H, "CHANNEL_MAJOR", cm)
        op = H.HyenaOperator(d_model=D, l_max=L + 2, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10,
                             lr=6e-4, wd=0.0, lr_pos_emb=0.0).to(dev)
        if sd is None:
            sd = {k_: v.clone() for k_, v in op.state_dict().items()}
        op.load_state_dict(sd)
        u = u0.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = op(u)
        y.backward(dy)
        res.append([y.float(), u.grad.float()] + [p.grad.float() for _, p in sorted(op.named_parameters())])
    for a, b_ in zip(*res):
        assert _rel(a, b_) < 2e-2, _rel(a, b_)
