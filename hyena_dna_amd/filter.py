"""The implicit Hyena filter as one autograd function over the fused HIP kernels of ``include/hyena_filter.h``.

Reference: ``HyenaFilter.filter`` (``src/models/sequence/hyena.py:229-238``) = positional embedding slice (130-131) ->
``Linear(E, 64) / Sin / Linear(64, 64) / Sin / Linear(64, 64) / Sin / Linear(64, D, bias=False)`` (199-215) ->
``ExponentialModulation`` (152-155), then the ``l d -> d l`` rearrange of ``HyenaOperator.forward`` (405-412).  The
function returns the filter directly as ``(D, L)`` fp32, the layout ``hyena_fftconv_*`` reads.

``hyena_dna_amd.hyena.HyenaFilter.filter_dl`` uses it whenever ``fused_filter_ok`` holds (the HyenaDNA configuration)
and takes its PyTorch path otherwise.

Precision follows the reference's graph.  Without autocast: fp32 throughout (``hyena_filter_fwd`` / ``_bwd``, exact-fp32 MFMA).
Under ``torch.autocast`` the reference's four ``nn.Linear`` run in the 16-bit autocast type -- inputs, weights, biases and outputs
rounded to it, fp32 accumulation -- while ``Sin`` and the modulation are promoted to fp32; ``hyena_filter16_fwd`` / ``_bwd``
(``csrc/filter16_kernels.h``) compute exactly that graph on the 16-bit matrix cores (the filter bit-identical to the oracle under CPU
autocast, ``tests/test_filter16_emu.py``).  ``HYENA_FILTER_AUTOCAST=fp32`` keeps the fp32 kernels under autocast (the behaviour up to
round 2: closer to the fp64 truth, but 1e-2 ... 2e-1 away from what the reference computes there).  What is kept for the backward: the
pre-activations of the three sine layers (3 x 64 x L, fp32 or 16-bit).
"""
import os

import torch

from . import _gradmode, _lib

__all__ = ["hyena_filter_dl", "HyenaFilterFunc", "fused_filter_ok"]


def fused_filter_ok(L, emb_dim, order, d_model, num_inner_mlps, normalized, linear_mixer):
    return (not normalized and not linear_mixer and num_inner_mlps == 2
            and _lib.filter_supported(L, emb_dim, order, d_model))


def _f32(x):
    return None if x is None else x.detach().to(torch.float32).contiguous()


class HyenaFilterFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, t, w0, b0, w1, b1, w2, b2, w3, freq, deltas, shift, modulate, compute_dtype=None):
        """z (L, E), t (L,), w0 (64, E), b0 (64,), w1/w2 (64, 64), b1/b2 (64,), w3 (D, 64), freq (64,), deltas (D,)
        -> k (D, L) fp32.  ``compute_dtype``: None (fp32 graph) or the autocast type whose graph is to be computed."""
        args = [_f32(x) for x in (z, t, w0, b0, w1, b1, w2, b2, w3, freq, deltas)]
        need = _gradmode.needs(ctx)
        want_grad = any(need)
        if any(need[i] for i in (1, 10)):
            raise NotImplementedError("gradients w.r.t. pos_emb.t / modulation.deltas are not provided by the fused "
                                      "filter kernels (both are buffers in every HyenaDNA configuration)")
        if want_grad:
            k, saved = _lib.filter_fwd(*args, shift, modulate, save=True, compute_dtype=compute_dtype)
            ctx.save_for_backward(saved, *args)
        else:
            k = _lib.filter_fwd(*args, shift, modulate, save=False, compute_dtype=compute_dtype)
        ctx.meta = (shift, modulate, [x.dtype for x in (z, w0, b0, w1, b1, w2, b2, w3, freq)], compute_dtype)
        return k

    @staticmethod
    def backward(ctx, dk):
        saved, *args = ctx.saved_tensors
        shift, modulate, dtypes, compute_dtype = ctx.meta
        g = _lib.filter_bwd(_lib.as_rows(dk.to(torch.float32)), saved, *args, shift, modulate, need_dz=ctx.needs_input_grad[0],
                            compute_dtype=compute_dtype)
        dw0, db0, dw1, db1, dw2, db2, dw3, dfreq, dz = g
        outs = [dz, dw0, db0, dw1, db1, dw2, db2, dw3, dfreq]
        outs = [None if o is None else o.to(dt) for o, dt in zip(outs, dtypes)]
        dz, dw0, db0, dw1, db1, dw2, db2, dw3, dfreq = outs
        return dz, None, dw0, db0, dw1, db1, dw2, db2, dw3, dfreq, None, None, None, None


def autocast_compute_dtype(device_type="cuda"):
    """The 16-bit type whose graph the filter computes right now: the autocast type of ``device_type`` if autocast is on there (and
    ``HYENA_FILTER_AUTOCAST`` is not ``fp32``), else None (fp32)."""
    if os.environ.get("HYENA_FILTER_AUTOCAST", "").lower() in ("fp32", "float32", "off", "0"):
        return None
    if not torch.is_autocast_enabled(device_type):
        return None
    dt = torch.get_autocast_dtype(device_type)
    return dt if dt in (torch.bfloat16, torch.float16) else None


def hyena_filter_dl(z, t, w0, b0, w1, b1, w2, b2, w3, freq, deltas, shift=0.0, modulate=True, compute_dtype="auto"):
    if compute_dtype == "auto":
        compute_dtype = autocast_compute_dtype(z.device.type)
    return _gradmode.apply(HyenaFilterFunc, z, t, w0, b0, w1, b1, w2, b2, w3, freq, deltas, shift, modulate, compute_dtype)
