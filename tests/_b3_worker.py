"""Worker of tests/test_overlay_reference.py::test_reference_python_op_runs_on_the_native_seam: the reference's OWN,
unmodified ``src/ops/fftconv.py`` (``FFTConvFunc``, lines 58-108) imported on top of this repository's ``fftconv`` module
(overlay/fftconv.py -> C ABI), against the reference's own ``fftconv_ref`` of the same file.  Kernels under tests/hipemu."""
import importlib.util
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("HYENA_REFERENCE", "/root/reference")


def main():
    sys.path[:0] = [os.path.join(ROOT, "overlay"), ROOT]
    from hyena_dna_amd import _lib
    from tests.hipemu.emu_backend import EmuBackend
    _lib._backend = EmuBackend()
    import fftconv as native
    assert os.path.realpath(native.__file__) == os.path.realpath(os.path.join(ROOT, "overlay", "fftconv.py"))
    # load the REFERENCE's python op by file path (the overlay also has a src/ops/fftconv.py: this test wants the original)
    spec = importlib.util.spec_from_file_location("ref_ops_fftconv", os.path.join(REF, "src", "ops", "fftconv.py"))
    ref_ops = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_ops)
    assert ref_ops.fftconv_fwd is native.fftconv_fwd and ref_ops.fftconv_bwd is native.fftconv_bwd

    worst = 0.0
    for (B, H, L, dtype) in [(2, 4, 128, torch.float32), (2, 3, 1000, torch.float32), (1, 2, 4100, torch.float32), (2, 3, 37, torch.float32),
                             (2, 4, 1024, torch.bfloat16)]:
        g = torch.Generator().manual_seed(L)
        u = torch.randn(B, H, L, generator=g).to(dtype)
        k = torch.randn(H, L, generator=g) * torch.exp(-5.0 * torch.linspace(0, 1, L))[None] * 0.1
        D = torch.randn(H, generator=g)
        dout = torch.randn(B, H, L, generator=g).to(dtype)
        res = []
        for fn in (lambda a, b, c: ref_ops.FFTConvFunc.apply(a, b, c, None, False),          # -> native seam (this repo)
                   lambda a, b, c: ref_ops.fftconv_ref(a, b, c, None, gelu=False)):          # the reference's torch.fft path
            a, b, c = u.clone().requires_grad_(True), k.clone().requires_grad_(True), D.clone().requires_grad_(True)
            y = fn(a, b, c)
            y.backward(dout)
            res.append((y.detach().float(), a.grad.float(), b.grad, c.grad))
        tol = 3e-6 if dtype == torch.float32 else 2e-2
        for x, r in zip(*res):
            e = ((x - r).norm() / r.norm().clamp_min(1e-30)).item()
            worst = max(worst, e) if dtype == torch.float32 else worst
            assert e < tol, (B, H, L, dtype, e)
    # the 14-argument native seam itself serves the plain form only: the reference's autograd class with gelu=True is refused there,
    # not mis-computed (fftconv_func of this package composes the H3-form options around the kernels: below)
    try:
        ref_ops.FFTConvFunc.apply(u, k, D, None, True)
        raise SystemExit("gelu=True should have raised")
    except NotImplementedError:
        pass
    # The H3 form and the element-wise options (round 4): the REFERENCE's own fftconv_h3_ref / fftconv_ref (src/ops/fftconv.py:15-55)
    # against this package's fftconv_func on the same kernels, and against the oracle's restatement (which this pins).
    from hyena_dna_amd.fftconv import fftconv_func
    from oracle import hyena_oracle as O
    rel = lambda x, r: ((x.double() - r.double()).norm() / r.double().norm().clamp_min(1e-30)).item()
    for head_dim in (1, 8):
        b, h, L = 2, 3, 300
        g = torch.Generator().manual_seed(40 + head_dim)
        kin, v, q = (torch.randn(b, h * head_dim, L, generator=g) for _ in range(3))
        ssm = torch.randn(h, L, generator=g) * torch.exp(-3.0 * torch.linspace(0, 1, L))[None] * 0.3
        rev = torch.randn(h, L, generator=g) * 0.05
        D = torch.randn(h, generator=g)
        for r_ in (None, rev):
            want = ref_ops.fftconv_h3_ref(kin, ssm, D, q, v, head_dim, r_)
            assert torch.equal(O.fftconv_h3_ref(kin, ssm, D, q, v, head_dim, r_), want)              # the restatement, bit for bit
            assert rel(fftconv_func(kin, ssm, D, gelu=False, v=v, head_dim=head_dim, q=q, k_rev=r_), want) < 1e-5
    mask = (torch.rand(2, 4, generator=g) > 0.4).float() * 1.5
    u4 = torch.randn(2, 4, 600, generator=g)
    k4 = torch.randn(4, 600, generator=g) * 0.05
    D4 = torch.randn(4, generator=g)
    for gelu in (True, False):
        want = ref_ops.fftconv_ref(u4, k4, D4, mask, gelu=gelu)
        assert torch.equal(O.fftconv_ref(u4, k4, D4, dropout_mask=mask, gelu=gelu), want)
        assert rel(fftconv_func(u4, k4, D4, dropout_mask=mask, gelu=gelu), want) < 3e-6
    print(f"B3_OK worst_rel_fp32={worst:.2e}", flush=True)


if __name__ == "__main__":
    main()
