"""INTEGRATION.md section 3, on hardware: the C++ torch-extension binding a maintainer would put in place of
csrc/fftconv/fftconv.cpp (integration/fftconv_binding.cpp: module `fftconv`, the reference's 14-argument fftconv_fwd /
fftconv_bwd over the C ABI) is compiled here and driven exactly as src/ops/fftconv.py:58-103 drives the reference's
(rfft of the filter before, irfft of dfilter after), against the oracle."""
import math
import os
import tempfile

import pytest
import torch

from oracle import hyena_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def binding(gpu_lib):
    from torch.utils.cpp_extension import load
    csrc = os.path.join(ROOT, "hyena_dna_amd", "csrc")
    return load(name="fftconv", sources=[os.path.join(ROOT, "integration", "fftconv_binding.cpp")],
                extra_include_paths=[os.path.join(ROOT, "include"), "/opt/rocm/include"],
                extra_cflags=["-D__HIP_PLATFORM_AMD__", "-O2"],
                extra_ldflags=["-L" + csrc, "-lhyena_fftconv", "-Wl,-rpath," + csrc], with_cuda=False, is_python_module=True,
                build_directory=tempfile.mkdtemp(prefix="fftconv_binding_"))


@pytest.mark.parametrize("B,H,L,dtype", [(2, 8, 1024, torch.float32), (2, 4, 3000, torch.float32), (1, 8, 40000, torch.float32),
                                          (2, 8, 4096, torch.bfloat16)])
def test_binding_as_the_reference_python_op_calls_it(binding, B, H, L, dtype):
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(L)
    u = torch.randn(B, H, L, generator=g).to(dtype)
    k = torch.randn(H, L, generator=g) * torch.exp(-5.0 * torch.linspace(0, 1, L))[None] * 0.1
    D = torch.randn(H, generator=g)
    dout = torch.randn(B, H, L, generator=g).to(dtype)
    fft_size = max(2 * 2 ** int(math.ceil(math.log2(L))), 16)                    # src/ops/fftconv.py:65
    k_f = torch.fft.rfft(k.to(dev), n=fft_size).contiguous()                      # src/ops/fftconv.py:66
    out = binding.fftconv_fwd(u.to(dev), k_f, D.to(dev), None, 1, None, None, False, False, False, fft_size, False, False, False)
    du, dk_f, dD, dv, dq = binding.fftconv_bwd(dout.to(dev), u.to(dev), k_f, D.to(dev), None, 1, None, None, False, False, False,
                                               fft_size, False, False)
    dk = torch.fft.irfft(dk_f, n=fft_size, norm="forward")[..., :L]              # src/ops/fftconv.py:98
    assert dv is None and dq is None
    u_, k_, D_ = u.clone().requires_grad_(True), k.clone().requires_grad_(True), D.clone().requires_grad_(True)
    r = O.fftconv_ref(u_, k_, D_)
    r.backward(dout)

    def rel(a, b):
        a, b = a.detach().double().cpu(), b.detach().double()
        return ((a - b).norm() / b.norm()).item()
    tol = 3e-6 if dtype == torch.float32 else 1.2e-2
    assert rel(out, r) < tol and rel(du, u_.grad) < tol
    assert rel(dk, k_.grad) < 3e-6 and rel(dD, D_.grad) < 1e-5
    with pytest.raises(RuntimeError):
        binding.fftconv_fwd(u.to(dev), k_f, D.to(dev), None, 1, None, None, True, False, False, fft_size, False, False, False)   # gelu


def test_reference_op_options_and_long_sequences_on_the_gpu(gpu_lib, monkeypatch):
    """fftconv_func's H3-form options and the split of sequences beyond the largest transform, on the gfx950 kernels, against the
    oracle's restatements of src/ops/fftconv.py:15-55 (the emulator runs the same cases in tests/test_fftconv_options.py)."""
    from hyena_dna_amd import _lib
    from hyena_dna_amd.fftconv import fftconv_func
    from oracle import hyena_oracle as O
    dev = torch.device("cuda", 0)
    rel = lambda a, b: ((a.double().cpu() - b.double()).norm() / b.double().norm()).item()
    g = torch.Generator().manual_seed(11)
    # GELU + dropout mask, values and gradients
    u, k, D = torch.randn(3, 8, 3000, generator=g), torch.randn(8, 3000, generator=g) * 0.05, torch.randn(8, generator=g)
    mask = (torch.rand(3, 8, generator=g) > 0.3).float() / 0.7
    dy = torch.randn(3, 8, 3000, generator=g)
    t = [x.clone().to(dev).requires_grad_(True) for x in (u, k, D)]
    out = fftconv_func(t[0], t[1], t[2], dropout_mask=mask.to(dev), gelu=True)
    out.backward(dy.to(dev))
    r = [x.clone().requires_grad_(True) for x in (u, k, D)]
    want = O.fftconv_ref(r[0], r[1], r[2], dropout_mask=mask, gelu=True)
    want.backward(dy)
    assert rel(out, want) < 3e-6 and all(rel(a.grad, b.grad) < 1e-5 for a, b in zip(t, r))
    # the H3 form, head_dim 8
    b, h, hd, L = 2, 4, 8, 2048
    kin, v, q = (torch.randn(b, h * hd, L, generator=g) for _ in range(3))
    ssm = torch.randn(h, L, generator=g) * 0.05
    Dh = torch.randn(h, generator=g)
    got = fftconv_func(kin.to(dev), ssm.to(dev), Dh.to(dev), gelu=False, v=v.to(dev), head_dim=hd, q=q.to(dev))
    assert rel(got, O.fftconv_h3_ref(kin, ssm, Dh, q, v, head_dim=hd)) < 1e-5
    # beyond the largest transform: the real limit (L = 2^20 + 4096, two channels) and a lowered one (both plans inside the split)
    for L, cap in ((2 ** 20 + 4096, None), (100001, 40000)):
        if cap is not None:
            monkeypatch.setattr(_lib, "MAX_L", cap)
        u = torch.randn(1, 2, L, generator=g)
        k = torch.randn(2, L, generator=g) * torch.exp(-6.0 * torch.linspace(0, 1, L))[None] * 0.1
        D2 = torch.randn(2, generator=g)
        dy = torch.randn(1, 2, L, generator=g)
        t = [x.clone().to(dev).requires_grad_(True) for x in (u, k, D2)]
        out = fftconv_func(t[0], t[1], t[2], gelu=False)
        out.backward(dy.to(dev))
        r = [x.clone().requires_grad_(True) for x in (u, k, D2)]
        want = O.fftconv_ref(r[0], r[1], r[2], gelu=False)
        want.backward(dy)
        assert rel(out, want) < 5e-6, L
        assert rel(t[0].grad, r[0].grad) < 5e-6 and rel(t[1].grad, r[1].grad) < 5e-6 and rel(t[2].grad, r[2].grad) < 2e-5, L
