"""One batched weight cast per optimizer step instead of one small cast kernel per weight and use (round 6).

Under ``torch.autocast`` every 16-bit product of this package needs its fp32 parameter in the compute type: ``in_proj`` / ``out_proj`` / ``fc1`` / ``fc2``
weights and biases, the LM head -- ~10 casts per layer forward and, through autograd's ``ToCopyBackward``, two more per weight in the backward (the
weight gradient is rounded to the 16-bit type and converted back for the fp32 ``.grad``): ~170 launches of 5 - 9 us per 8-layer step at L = 2^20
(profiles/r6a_copies_model.txt), ~1 ms of a 155 ms step and far more of a launch-bound one.

``shadow(p, dtype)`` is a 16-bit copy of parameter ``p`` that is refreshed for ALL registered parameters at once -- one ``torch._foreach_copy_`` -- the
first time any of them is asked for after a change (the version counter of a tensor moves with every in-place update: optimizer steps,
``load_state_dict``, ``copy_``; a re-allocated parameter shows in its data pointer).  Values: exactly ``p.detach().to(dtype)``.  ``rounded_f32(p, dtype)``:
those values back in fp32 (what a bias rounded "as autocast rounds it" is added as), refreshed in the same pass.

The autograd functions of this package (projection.py, mixer.py, lm.FusedMlpFunc) take the fp32 PARAMETER and look its shadow up in their forward;
their backward returns the weight gradient in the parameter's own type -- fp32, as the weight-gradient products accumulate it: no conversion kernels
in the backward either.  ``ROUND_WGRAD`` (HYENA_WGRAD_ROUND16=1) rounds it through the 16-bit type first, as autocast's own graph does (the reference's
semantics to the bit of that rounding; two more small kernels per weight): the default differs from the autocast graph by that one 2^-9 rounding per
gradient element, on the accurate side.

HYENA_CAST_CACHE=0: plain ``p.to(dtype)`` per use and the rounding of the weight gradients, as rounds 1 - 5 had it.
"""
import os
import weakref

import torch

ENABLED = os.environ.get("HYENA_CAST_CACHE", "1") != "0"
ROUND_WGRAD = os.environ.get("HYENA_WGRAD_ROUND16", "0") == "1" or not ENABLED

__all__ = ["shadow", "rounded_f32", "wgrad_out", "reset", "invalidate", "stats"]


class _Entry:
    __slots__ = ("ref", "shadow", "f32", "version", "ptr")

    def __init__(self, p):
        self.ref = weakref.ref(p)
        self.shadow = None
        self.f32 = None          # allocated on first use of rounded_f32
        self.version = -1
        self.ptr = 0


_entries = {}          # (id(param), dtype) -> _Entry
_counters = {"bulk_refreshes": 0, "tensors_refreshed": 0, "hits": 0}


def reset():
    """drop every shadow (entries also die with their parameter; this only frees them early)"""
    _entries.clear()


def invalidate():
    """mark every shadow stale: the next use refreshes them all in one pass (lm.GraphedTrainStep calls this right before its capture, so that the
    refresh is PART of the captured step -- a shadow that happened to be fresh at capture time would never be refreshed by the replays)"""
    for e in _entries.values():
        e.version = -1


def stats():
    return dict(_counters, entries=len(_entries))


def _stale(e, p):
    return (e.shadow is None or e.version != p._version or e.ptr != p.data_ptr() or e.shadow.shape != p.shape or e.shadow.device != p.device)


def _refresh(dtype, device):
    """every stale shadow of this (dtype, device) in ONE multi-tensor copy (+ one for the fp32 images of the rounded values)"""
    dst, src, ents, dead, f_dst, f_src = [], [], [], [], [], []
    for key, e in _entries.items():
        p = e.ref()
        if p is None:
            dead.append(key)
            continue
        if key[1] != dtype or p.device != device or not _stale(e, p):
            continue
        if e.shadow is None or e.shadow.shape != p.shape or e.shadow.device != p.device:
            e.shadow = torch.empty_like(p, dtype=dtype, memory_format=torch.contiguous_format)
            if e.f32 is not None:
                e.f32 = torch.empty_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
        dst.append(e.shadow)
        src.append(p.detach())
        ents.append((e, p))
        if e.f32 is not None:
            f_dst.append(e.f32)
            f_src.append(e.shadow)
    for key in dead:
        del _entries[key]
    if dst:
        with torch.no_grad():
            torch._foreach_copy_(dst, src)
            if f_dst:
                torch._foreach_copy_(f_dst, f_src)
        for e, p in ents:
            e.version, e.ptr = p._version, p.data_ptr()
        _counters["bulk_refreshes"] += 1
        _counters["tensors_refreshed"] += len(dst)


def _cacheable(p, dtype):
    return (ENABLED and torch.is_tensor(p) and p.dtype == torch.float32 and dtype in (torch.bfloat16, torch.float16) and p.is_leaf and p.dim() > 0
            and isinstance(p, torch.nn.Parameter))


def _entry(p, dtype):
    key = (id(p), dtype)
    e = _entries.get(key)
    if e is None or e.ref() is not p:
        e = _entries[key] = _Entry(p)
    return e


def shadow(p, dtype):
    """``p.detach().to(dtype).contiguous()`` -- for an fp32 nn.Parameter and a 16-bit dtype from the per-step shadow (never write to the result)"""
    if p is None:
        return None
    if not _cacheable(p, dtype):
        return p.detach().to(dtype).contiguous()
    e = _entry(p, dtype)
    if _stale(e, p):
        _refresh(dtype, p.device)
    else:
        _counters["hits"] += 1
    return e.shadow


def rounded_f32(p, dtype):
    """``p.detach().to(dtype).to(torch.float32)``: the parameter's values rounded to the compute type, in fp32 (how the kernels take a bias)"""
    if p is None:
        return None
    if not _cacheable(p, dtype):
        return p.detach().to(dtype).to(torch.float32).contiguous()
    e = _entry(p, dtype)
    if e.f32 is None:
        e.f32 = torch.empty_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
        e.version = -1                                   # (fill it in the next pass)
    if _stale(e, p):
        _refresh(dtype, p.device)
    else:
        _counters["hits"] += 1
    return e.f32


def wgrad_out(g, param_dtype, compute_dtype):
    """the gradient an autograd function hands back for a parameter of type ``param_dtype`` that it used in ``compute_dtype``: ``g`` (an fp32 sum) in the parameter's type;
    with ROUND_WGRAD through the compute type first (autocast's graph: the gradient of a 16-bit weight exists in 16 bits before it reaches the fp32 leaf)"""
    if g is None:
        return None
    if ROUND_WGRAD and compute_dtype in (torch.bfloat16, torch.float16) and g.dtype != compute_dtype:
        g = g.to(compute_dtype)
    return g if g.dtype == param_dtype else g.to(param_dtype)
