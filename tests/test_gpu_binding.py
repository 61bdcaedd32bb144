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
