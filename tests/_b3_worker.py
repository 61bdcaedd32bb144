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
    # options outside the HyenaDNA path are refused, not mis-computed
    try:
        ref_ops.FFTConvFunc.apply(u, k, D, None, True)
        raise SystemExit("gelu=True should have raised")
    except NotImplementedError:
        pass
    print(f"B3_OK worst_rel_fp32={worst:.2e}", flush=True)


if __name__ == "__main__":
    main()
