"""The one-launch-per-direction kernels for short rows with a small batch (csrc/onchip_kernels.h: small_fwd_kernel /
small_bwd_kernel; hyenadna-tiny-1k: L = 1024, B = 8, D = 128) on a real MI355X through the C ABI: against the oracle, against the
general kernels of the same library (HYENA_FFTCONV_SMALL=0), options and repetitions bitwise."""
import pytest
import torch

from oracle import hyena_oracle as O

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,D,L", [(8, 128, 1024), (8, 256, 1024), (3, 37, 1000), (16, 5, 700), (1, 2, 37), (5, 130, 2048), (8, 64, 1500),
                                   (7, 3, 1025)])
def test_small_fused_pair_on_gpu(gpu_lib, monkeypatch, B, D, L, dtype):
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(3 * L + B)
    u = torch.randn(B, D, L, generator=g).to(dtype)
    k = torch.randn(D, L, generator=g) * torch.exp(-5.0 * torch.linspace(0, 1, L))[None] * 0.1
    bias = torch.randn(D, generator=g)
    dout = torch.randn(B, D, L, generator=g).to(dtype)
    ud, kd, bd, dd = (t.to(dev) for t in (u, k, bias, dout))
    out, saved = gpu_lib.fftconv_fwd(ud, kd, bd, save=True)
    du, dk, dbias = gpu_lib.fftconv_bwd(dd, ud, kd, bd, saved=saved)
    assert torch.equal(out, gpu_lib.fftconv_fwd(ud, kd, bd))
    monkeypatch.setenv("HYENA_FFTCONV_SMALL", "0")
    g_out, g_saved = gpu_lib.fftconv_fwd(ud, kd, bd, save=True)
    g_du, g_dk, g_db = gpu_lib.fftconv_bwd(dd, ud, kd, bd, saved=g_saved)
    monkeypatch.delenv("HYENA_FFTCONV_SMALL")
    # two kernels, the same transform: on the GPU the compiler contracts their products into FMAs differently, so the comparison
    # is to rounding noise, not bitwise (under tests/hipemu, where nothing is contracted, the two are bit-identical)
    tol_x = 1e-6 if dtype == torch.float32 else 1e-4        # 16-bit outputs: a handful of neighbouring-value roundings
    assert _rel(out.float(), g_out.float()) < tol_x and _rel(du.float(), g_du.float()) < tol_x and _rel(dk, g_dk) < 1e-6
    if dtype != torch.float32:
        assert (out != g_out).float().mean().item() < 0.02 and (du != g_du).float().mean().item() < 0.02
    # oracle on the same (16-bit) inputs in fp32
    u_, k_, b_ = u.float().requires_grad_(True), k.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    r_out = O.fftconv_ref(u_, k_, b_)
    r_out.backward(dout.float())
    tol = 2e-6 if dtype == torch.float32 else (2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10)
    assert _rel(out.float(), r_out) < tol and _rel(du.float(), u_.grad) < tol and _rel(dk, k_.grad) < 3e-6
    db64 = (dout.double() * u.double()).sum(dim=(0, 2))
    assert (dbias.double().cpu() - db64).abs().max() < 1e-6 * (B * L) ** 0.5 + 1e-6
    du2, dk2, db2 = gpu_lib.fftconv_bwd(dd, ud, kd, bd, need_du=True, need_dk=False, saved=saved)
    assert dk2 is None and torch.equal(du2, du)
    du3, dk3, db3 = gpu_lib.fftconv_bwd(dd, ud, kd, bd, need_du=False, need_dk=True, saved=saved)
    assert du3 is None and torch.equal(dk3, dk) and torch.equal(db3, dbias)
    for _ in range(3):
        du4, dk4, db4 = gpu_lib.fftconv_bwd(dd, ud, kd, bd, saved=saved)
        assert torch.equal(du4, du) and torch.equal(dk4, dk) and torch.equal(db4, dbias)
    du5, dk5, db5 = gpu_lib.fftconv_bwd(dd, ud, kd, bd)             # no saved spectrum: the same kernels transform the filter first
    assert torch.equal(du5, du) and torch.equal(dk5, dk) and torch.equal(db5, dbias)
