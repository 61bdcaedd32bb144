"""``fftconv`` -- the module the UNMODIFIED reference ``src/ops/fftconv.py`` imports its native entry points from
(``from fftconv import fftconv_fwd, fftconv_bwd``, src/ops/fftconv.py:8), with the reference binding's exact call
signatures (csrc/fftconv/fftconv.cpp:53-61 and 134-142, pybind11 module ``fftconv``, fftconv.cpp:238-241), implemented on
the C ABI of libhyena_fftconv.so (include/hyena_fftconv.h) through ``hyena_dna_amd._lib``.

Put ``<repo>/overlay`` on ``sys.path`` and the reference's own autograd function
``src.ops.fftconv.FFTConvFunc`` -- unmodified, including its ``torch.fft.rfft(k)`` before and ``torch.fft.irfft(dk_f)``
after the native calls -- runs its long convolutions on the MI355X kernels.  (The overlay ALSO ships
``overlay/src/ops/fftconv.py``, which replaces that Python layer altogether and skips the two extra FFTs; this module is
the seam for a tree that keeps its own ``src/ops/fftconv.py``.)

Interface translation (the reference's native seam works in the frequency domain, this library in the time domain):
  * ``filter`` = rfft(k, n=fft_size), complex64 (H, fft_size/2 + 1)  ->  k = irfft(filter, n=fft_size)[..., :L] on the device;
  * ``dfilter`` must satisfy irfft(dfilter, n=fft_size, norm='forward')[..., :L] = dk (src/ops/fftconv.py:98)
    ->  dfilter = rfft(dk zero-padded to fft_size, norm='forward');  the library already sums dk / dD over the batch
    (the reference kernel returns per-batch partials and its binding sums them, fftconv.cpp:209-210,235).
Lifted restrictions: any L (the reference kernel needs even L and fft_size <= 16384, fftconv.cpp:114-115).  Options no
HyenaDNA configuration enables (gelu, dropout_mask, head_dim = 8, q / v, output_hbl_layout, fftfp16, gelu_inp / gelu_q)
raise instead of silently computing something else.
"""
import torch

from hyena_dna_amd import _lib

__all__ = ["fftconv_fwd", "fftconv_bwd"]


def _check(u, filt, D, v, head_dim, q, dropout_mask, gelu, gelu_inp, gelu_q, fft_size, output_hbl_layout, fftfp16):
    bad = [n for n, on in (("gelu", gelu), ("gelu_inp", gelu_inp), ("gelu_q", gelu_q), ("v", v is not None), ("q", q is not None),
                           ("dropout_mask", dropout_mask is not None), ("head_dim != 1", head_dim != 1),
                           ("output_hbl_layout", output_hbl_layout), ("fftfp16", fftfp16)) if on]
    if bad:
        raise NotImplementedError("fftconv (MI355X): option(s) " + ", ".join(bad) + " are not used by any HyenaDNA configuration")
    B, H, L = u.shape
    if u.stride(-1) != 1 or not filt.is_contiguous() or not D.is_contiguous():
        raise ValueError("fftconv: u must have unit stride along L; filter and D must be contiguous (fftconv.cpp:66-68)")
    if tuple(filt.shape) != (H, fft_size // 2 + 1) or tuple(D.shape) != (H,):
        raise ValueError(f"fftconv: expected filter (H, fft_size/2+1) = ({H}, {fft_size // 2 + 1}) and D ({H},), got "
                         f"{tuple(filt.shape)} and {tuple(D.shape)}")
    if filt.dtype != torch.complex64 or D.dtype != torch.float32:
        raise TypeError("fftconv: filter must be complex64 and D float32 (fftconv.cpp:79-81)")
    if L > fft_size // 2:
        raise ValueError("fftconv: L must be <= fft_size / 2 (fftconv.cpp:114)")
    return B, H, L


def _time_domain_filter(filt, fft_size, L):
    return torch.fft.irfft(filt, n=fft_size)[..., :L].contiguous()


def fftconv_fwd(u, filter, D, v, head_dim, q, dropout_mask, gelu, gelu_inp, gelu_q, fft_size, force_fp16_output,
                output_hbl_layout, fftfp16):
    """csrc/fftconv/fftconv.cpp:53-132."""
    B, H, L = _check(u, filter, D, v, head_dim, q, dropout_mask, gelu, gelu_inp, gelu_q, fft_size, output_hbl_layout, fftfp16)
    out = _lib.fftconv_fwd(u.contiguous(), _time_domain_filter(filter, fft_size, L), D)
    if force_fp16_output and u.dtype == torch.float32:            # (ignored for bf16 inputs, fftconv.cpp:109)
        out = out.to(torch.float16)
    return out


def fftconv_bwd(dout, u, filter, D, v, head_dim, q, dropout_mask, gelu, gelu_inp, gelu_q, fft_size, output_hbl_layout, fftfp16):
    """csrc/fftconv/fftconv.cpp:134-236: returns (du, dfilter, dD, dv, dq)."""
    B, H, L = _check(u, filter, D, v, head_dim, q, dropout_mask, gelu, gelu_inp, gelu_q, fft_size, output_hbl_layout, fftfp16)
    k = _time_domain_filter(filter, fft_size, L)
    du, dk, dD = _lib.fftconv_bwd(dout.to(u.dtype).contiguous(), u.contiguous(), k, D)
    dfilter = torch.fft.rfft(dk, n=fft_size, norm="forward")
    return du, dfilter, dD, None, None
