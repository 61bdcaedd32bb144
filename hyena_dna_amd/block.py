"""The residual glue of a HyenaDNA block -- (dropout ->) add -> LayerNorm -- on the fused HIP kernels of
``include/hyena_block.h``, behind the name and signature the reference imports for exactly this
(``src/models/sequence/long_conv_lm.py:31``: ``from flash_attn.ops.layer_norm import dropout_add_layer_norm``; used at
``long_conv_lm.py:387-396`` and, inside flash_attn's ``Block``, for the two norms of every layer --
restated unfused at ``src/models/sequence/simple_lm.py:267-271, 280-284``).

    out            = LayerNorm(dropout(x0) + residual)                          prenorm=False
    out, residual' = LayerNorm(dropout(x0) + residual), dropout(x0) + residual   prenorm=True

``residual'`` is fp32 (``residual_in_fp32=True``, what HyenaDNA trains with); ``out`` has ``x0``'s dtype, computed in
fp32 and rounded once.  Dropout (p > 0: HyenaDNA's embedding dropout, which the reference applies as the first block's) is part of
the same pass since round 4: a 64-bit seed is drawn from PyTorch's generator of the device per call (so ``torch.manual_seed`` and graph
capture behave as with ``F.dropout``), the keep / drop decision of an element is a pure function of (seed, index) and the backward
regenerates it -- no mask tensor, no extra pass (csrc/block_kernels.h); the draws are NOT PyTorch's.  Shapes outside the kernels' coverage (D not a multiple of 64, or > 1024) take the same graph in PyTorch ops on
the same device; host tensors are refused (``HyenaLibraryError``), as is a missing library.
"""
import torch
import torch.nn.functional as F

from . import _lib

__all__ = ["dropout_add_layer_norm", "AddLayerNormFunc", "embedding_dropout_add_layer_norm", "embedding_fusable"]


class AddLayerNormFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, residual, weight, bias, eps, prenorm, dropout_p=0.0, seed=None):
        shape = x0.shape
        D = shape[-1]
        x2 = x0.reshape(-1, D).contiguous()
        r2 = None if residual is None else residual.reshape(-1, D).to(torch.float32).contiguous()
        w = weight.detach().to(torch.float32).contiguous()
        b = bias.detach().to(torch.float32).contiguous()
        out, res_out, mean, rstd = _lib.add_norm_fwd(x2, r2, w, b, eps, x0.dtype, dropout_p=dropout_p, seed=seed)
        ctx.save_for_backward(res_out, w, mean, rstd)
        ctx.drop = (float(dropout_p), seed)
        ctx.meta = (shape, x0.dtype, None if residual is None else residual.dtype, weight.dtype, bias.dtype, prenorm)
        ctx.mark_non_differentiable()
        if prenorm:
            return out.view(shape), res_out.view(shape)
        return out.view(shape)

    @staticmethod
    def backward(ctx, dout, *rest):
        res_out, w, mean, rstd = ctx.saved_tensors
        shape, x_dtype, r_dtype, w_dtype, b_dtype, prenorm = ctx.meta
        D = shape[-1]
        dres_out = rest[0] if prenorm and rest and rest[0] is not None else None
        d2 = dout.reshape(-1, D).contiguous()
        h2 = None if dres_out is None else dres_out.reshape(-1, D).to(torch.float32).contiguous()
        dx, dres, dw, db = _lib.add_norm_bwd(d2, h2, res_out, w, mean, rstd, x_dtype, need_dres=r_dtype is not None,
                                             dropout_p=ctx.drop[0], seed=ctx.drop[1])
        return (dx.view(shape), None if dres is None else dres.view(shape).to(r_dtype), dw.to(w_dtype), db.to(b_dtype),
                None, None, None, None)


class EmbedAddLayerNormFunc(torch.autograd.Function):
    """(out, residual') of the FIRST block's norm with the token embedding gathered inside the pass: residual' = dropout(table[ids]),
    out = LayerNorm(residual').  The (rows, D) embedding and its gradient are never materialised: the backward returns the table's
    gradient as per-token-class sums (include/hyena_block.h, hyena_embed_add_norm_*)."""

    @staticmethod
    def forward(ctx, ids, table, weight, bias, eps, out_dtype, dropout_p=0.0, seed=None):
        shape = tuple(ids.shape) + (table.shape[1],)
        i2 = ids.reshape(-1).contiguous()
        t = table.detach().to(torch.float32).contiguous()
        w = weight.detach().to(torch.float32).contiguous()
        b = bias.detach().to(torch.float32).contiguous()
        out, res_out, mean, rstd = _lib.embed_add_norm_fwd(i2, t, w, b, eps, out_dtype, dropout_p=dropout_p, seed=seed)
        ctx.save_for_backward(res_out, w, mean, rstd, i2)
        ctx.drop = (float(dropout_p), seed)
        ctx.meta = (shape, table.shape[0], table.dtype, weight.dtype, bias.dtype)
        return out.view(shape), res_out.view(shape)

    @staticmethod
    def backward(ctx, dout, dres_out):
        res_out, w, mean, rstd, i2 = ctx.saved_tensors
        shape, V, t_dtype, w_dtype, b_dtype = ctx.meta
        D = shape[-1]
        d2 = dout.reshape(-1, D).contiguous()
        h2 = None if dres_out is None else dres_out.reshape(-1, D).to(torch.float32).contiguous()
        dt, dw, db = _lib.embed_add_norm_bwd(d2, h2, res_out, i2, V, w, mean, rstd, dropout_p=ctx.drop[0], seed=ctx.drop[1])
        return None, dt.to(t_dtype), dw.to(w_dtype), db.to(b_dtype), None, None, None, None


def embedding_fusable(ids, embedding, norm_weight):
    """the conditions under which the first block's norm can gather the token embedding itself: a plain nn.Embedding of at most 16 fp32
    rows (the DNA vocabulary), d_model 64 / 128 / 256, ids on the kernels' device"""
    w = embedding.weight
    _lib._require_gpu(w, "embedding weight")
    return (ids.dtype == torch.int64 and ids.device == w.device and w.dtype == torch.float32 and embedding.padding_idx is None
            and embedding.max_norm is None and not embedding.sparse and norm_weight is not None
            and _lib.embed_add_norm_supported(w.shape[0], w.shape[1], torch.float32))


def embedding_dropout_add_layer_norm(ids, table, weight, bias, dropout_p, epsilon, out_dtype=None):
    """``dropout_add_layer_norm(F.embedding(ids, table), None, weight, bias, dropout_p, epsilon, prenorm=True, residual_in_fp32=True)`` in one
    pass (the caller checks ``embedding_fusable``).  ``out_dtype``: fp32 like the unfused call's, or the autocast type the next module would
    round it to anyway (one rounding of the fp32 LayerNorm result either way)."""
    if out_dtype is None:
        out_dtype = torch.float32
    if dropout_p > 0.0:
        if not dropout_p < 1.0:
            raise ValueError(f"dropout probability has to be in [0, 1), got {dropout_p}")
        seed = torch.empty(1, dtype=torch.int64, device=table.device).random_()
        return EmbedAddLayerNormFunc.apply(ids, table, weight, bias, epsilon, out_dtype, float(dropout_p), seed)
    return EmbedAddLayerNormFunc.apply(ids, table, weight, bias, epsilon, out_dtype)


def _fused_ok(x0, residual, weight):
    _lib._require_gpu(x0, "x0")           # host tensors are refused like everywhere else in this package: no CPU fallback
    if residual is not None and residual.shape != x0.shape:
        return False
    return weight is not None and _lib.add_norm_supported(x0.shape[-1], x0.dtype, x0.dtype)


def dropout_add_layer_norm(x0, residual, weight, bias, dropout_p, epsilon, rowscale=None, layerscale=None, prenorm=False,
                           residual_in_fp32=False, return_dropout_mask=False):
    """Drop-in for ``flash_attn.ops.layer_norm.dropout_add_layer_norm`` as the reference calls it."""
    if rowscale is not None or layerscale is not None or return_dropout_mask:
        raise NotImplementedError("rowscale / layerscale / return_dropout_mask are not used by any HyenaDNA configuration")
    if residual_in_fp32 and _fused_ok(x0, residual, weight):
        if dropout_p > 0.0:                               # (the reference passes p = 0 in eval mode, long_conv_lm.py:392)
            if not dropout_p < 1.0:
                raise ValueError(f"dropout probability has to be in [0, 1), got {dropout_p}")
            seed = torch.empty(1, dtype=torch.int64, device=x0.device).random_()      # from the device's generator, on the device
            return AddLayerNormFunc.apply(x0, residual, weight, bias, epsilon, prenorm, float(dropout_p), seed)
        return AddLayerNormFunc.apply(x0, residual, weight, bias, epsilon, prenorm)
    if dropout_p > 0.0:
        x0 = F.dropout(x0, dropout_p, training=True)
    # generic graph (simple_lm.py:267-271)
    res = x0 + residual if residual is not None else x0
    if residual_in_fp32:
        res = res.to(torch.float32)
    out = F.layer_norm(res.to(weight.dtype), (x0.shape[-1],), weight, bias, epsilon).to(x0.dtype)
    return (out, res) if prenorm else out
