"""The caller's grad mode, made visible to autograd.Function.forward.

``ctx.needs_input_grad`` reports the inputs' ``requires_grad`` flags whatever the grad mode: under ``torch.no_grad()`` (how a trained model
is served) it still says True for every parameter, and a forward that sizes its work on it keeps state no backward will ever read -- for the
long convolution 2 GB of column spectra per call at L = 2^20.  The package's wrappers call ``apply`` below instead of ``Func.apply``;
``needs(ctx)`` in a forward is ``ctx.needs_input_grad`` masked by the mode the wrapper was called in.
"""
import threading

import torch

_tls = threading.local()


def apply(func, *args):
    prev = getattr(_tls, "on", None)
    _tls.on = torch.is_grad_enabled()
    try:
        return func.apply(*args)
    finally:
        _tls.on = prev


def needs(ctx):
    on = getattr(_tls, "on", None)
    if on is None:                         # Func.apply called directly: no information, keep autograd's answer
        return tuple(ctx.needs_input_grad)
    return tuple(bool(n) and on for n in ctx.needs_input_grad)
