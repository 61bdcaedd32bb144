"""Model glue around the Hyena mixer, MI355X-side: what the reference's backbone imports from ``flash_attn``
(``src/models/sequence/long_conv_lm.py:18-33``) and, on top of it, a Lightning-free language model of the same structure.

The reference's ``ConvLMHeadModel`` (``long_conv_lm.py:400-502``) is built from flash_attn's ``GPT2Embeddings``,
``Block``, ``Mlp``, ``MHA`` and ``GenerationMixin`` -- CUDA-only in the pinned flash_attn, and restated in plain PyTorch
by the reference itself at ``src/models/sequence/simple_lm.py:26-305`` / ``standalone_hyenadna.py:302-561``.  The classes
below follow those restatements' semantics (same constructor arguments, attribute / state-dict names and forward
contracts), so that the unmodified reference backbone builds on ROCm once ``overlay/flash_attn`` (thin re-exports of
this module) is on ``sys.path`` -- INTEGRATION.md section 4.  What is MI355X-specific:

* ``Block`` routes its two (dropout ->) add -> LayerNorm steps through the fused HIP kernels of
  ``hyena_dna_amd.block.dropout_add_layer_norm`` when ``fused_dropout_add_ln=True``;
* ``Mlp`` (tanh-GELU, ``long_conv_lm.py:117-123``) runs its two products that contract over d_model on this package's
  weights-stationary MFMA kernels with the element-wise passes in their epilogues (``FusedMlpFunc``: fc1 + bias + GELU forward,
  (dy W2) * GELU' + bias-gradient sums backward; ``csrc/proj_kernels.h``); fc2, the input gradient and the weight gradients
  stay library GEMMs, the latter with the split-K schedule of ``hyena_dna_amd.projection``;
* the mixer is whatever ``mixer_cls`` builds -- ``hyena_dna_amd.hyena.HyenaOperator`` through the registry swap.

``HyenaDNALM`` = embedding -> n_layer x Block -> (dropout, add,) LayerNorm -> tied LM head, the hyenadna-* model family
(``hg38_hyena.yaml``), with ``loss()`` = next-token cross entropy: the full-model step ``bench.py`` times as a secondary
figure (north_star configuration 5).
"""
import math
import os
from collections import namedtuple
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from .block import dropout_add_layer_norm, embedding_dropout_add_layer_norm, embedding_fusable
from .projection import hyena_linear

__all__ = ["Mlp", "Block", "GPT2Embeddings", "MHA", "GenerationMixin", "HyenaDNALM", "sync_shared_params", "all_gather_raw"]


# ---------------------------------------------------------------------------------------------------------------------
# flash_attn.modules.mlp.Mlp  (semantics: simple_lm.py:192-212)
# ---------------------------------------------------------------------------------------------------------------------
class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, activation=F.gelu, return_residual=False,
                 device=None, dtype=None):
        factory_kwargs = {"device": device, "dtype": dtype}
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.return_residual = return_residual
        self.fc1 = nn.Linear(in_features, hidden_features, **factory_kwargs)
        self.activation = activation
        self.fc2 = nn.Linear(hidden_features, out_features, **factory_kwargs)

    def forward(self, x):
        if self._fused_ok(x):
            y = fused_mlp(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)
        else:
            y = hyena_linear(x, self.fc1.weight, self.fc1.bias)
            y = self.activation(y)
            y = hyena_linear(y, self.fc2.weight, self.fc2.bias)
        return y if not self.return_residual else (y, x)

    def _fused_ok(self, x):
        """The matrix-core MLP kernels (csrc/proj_kernels.h) cover the HyenaDNA block: tanh-GELU, 16-bit compute (autocast or
        plain), d_model 128 / 256, d_inner a multiple of 256, biases present."""
        act = self.activation
        tanh_gelu = (isinstance(act, partial) and act.func is F.gelu and act.keywords.get("approximate") == "tanh" and not act.args)
        if not (FUSED_MLP and tanh_gelu and self.fc1.bias is not None and self.fc2.bias is not None):
            return False
        from . import _lib
        if not (x.is_cuda or _lib._backend.name != "hip"):
            return False
        dt = _mlp_dtype(x, self.fc1.weight)
        rows = x.numel() // max(x.shape[-1], 1)
        # fc1's kernel contracts over in_features, the fused backward (`mlp_dh_dgelu_bwd`) over fc2.out_features: both must be a width the
        # kernels are built for, whatever `out_features` the caller chose (the reference's Mlp accepts any; other widths take the library path)
        return (dt is not None and rows >= 1 and self.fc2.in_features == self.fc1.out_features
                and _lib.mlp_supported(rows, self.fc1.in_features, self.fc1.out_features, dt)
                and _lib.mlp_supported(rows, self.fc2.out_features, self.fc1.out_features, dt))


FUSED_MLP = os.environ.get("HYENA_FUSED_MLP", "1") != "0"          # A/B knob: 0 = two library GEMMs + PyTorch's GELU
# HyenaDNALM pads batches of several odd-length sequences to a multiple of 64 positions (HyenaDNALM._aligned_length); 0 = run them as they come
PAD_SEQUENCES = os.environ.get("HYENA_LM_PAD_SEQUENCES", "1") != "0"
_SEQ_ALIGN = 64
_PAD_SINGLE_MIN = int(os.environ.get("HYENA_LM_PAD_SINGLE_MIN", "8192"))      # a single sequence is padded from this length on (_aligned_length) ...
_PAD_SINGLE_ROWS = 4096                                                        # ... if the padded length is a multiple of this (64 weight-gradient slices of 64-aligned rows)


def _mlp_dtype(x, weight):
    if torch.is_autocast_enabled():
        dt = torch.get_autocast_dtype("cuda" if x.is_cuda else "cpu")
        return dt if dt in (torch.bfloat16, torch.float16) else None
    return x.dtype if x.dtype in (torch.bfloat16, torch.float16) and weight.dtype == x.dtype else None


class FusedMlpFunc(torch.autograd.Function):
    """y = fc2(gelu_tanh(fc1(x))) (simple_lm.py:207-211) with the element-wise passes inside this package's MFMA kernels:

        forward   a, h = mlp_fc1_gelu_fwd(x, W1, b1)          one launch: a = x W1^T + b1 and h = gelu(a), both kept (a for the
                  y = h W2^T + b2                              backward, h as fc2's operand -- what autograd keeps today as well)
        backward  da, db1 = mlp_dh_dgelu_bwd(dy, W2^T, a)      one launch: (dy W2) * gelu'(a) and its column sums
                  dx = da W1;  dW1 = da^T x;  dW2 = dy^T h;  db2 = sum dy        library GEMMs (contractions over d_inner / positions)

    Gone: the GELU pass (read a, write h), the GELU-backward pass (read a, dh, write da), the dh tensor and fc1's bias-gradient
    reduction -- 12 of the ~28 GB a layer's MLP moves at L = 2^20."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, dt=None):
        # dt: autocast's compute type -- w1 ... b2 are then the fp32 PARAMETERS, used through their per-step shadows (_castcache: one batched cast
        # per optimizer step), and their gradients go back in fp32; None: operands already in the compute type
        from . import _castcache, _lib
        ctx.ptypes = (w1.dtype, b1.dtype, w2.dtype, b2.dtype, dt)
        if dt is not None:
            b1f = _castcache.rounded_f32(b1, dt)                             # b1 rounded to the compute type (autocast semantics), in fp32
            w1, w2, b2 = _castcache.shadow(w1, dt), _castcache.shadow(w2, dt), _castcache.shadow(b2, dt)
        else:
            b1f = b1.float().contiguous()
        x2 = x.reshape(-1, x.shape[-1])
        a, h = _lib.mlp_fc1_gelu_fwd(x2, w1, b1f)
        y = torch.addmm(b2, h, w2.t())
        ctx.save_for_backward(x2, w1, w2, a, h)
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], w2.shape[0])

    @staticmethod
    def backward(ctx, dy):
        from . import _castcache, _lib
        from .projection import split_k_weight_grad
        x2, w1, w2, a, h = ctx.saved_tensors
        t_w1, t_b1, t_w2, t_b2, dt = ctx.ptypes
        cdt = dt if dt is not None else dy.dtype
        dy2 = dy.reshape(-1, dy.shape[-1]).contiguous()
        da, db1 = _lib.mlp_dh_dgelu_bwd(dy2, w2.t().contiguous(), a)
        dx = dw1 = dw2 = db2 = None
        if ctx.needs_input_grad[0]:
            dx = torch.mm(da, w1).view(ctx.xshape)
        if ctx.needs_input_grad[1]:
            dw1 = _castcache.wgrad_out(split_k_weight_grad(da, x2), t_w1, cdt)
        if ctx.needs_input_grad[3]:
            dw2 = _castcache.wgrad_out(split_k_weight_grad(dy2, h), t_w2, cdt)
        if ctx.needs_input_grad[4]:
            db2 = _castcache.wgrad_out(_lib.colsum(dy2), t_b2, cdt)
        return dx, dw1, _castcache.wgrad_out(db1, t_b1, cdt) if ctx.needs_input_grad[2] else None, dw2, db2, None


def fused_mlp(x, w1, b1, w2, b2):
    dt = _mlp_dtype(x, w1)
    with torch.autocast("cuda" if x.is_cuda else "cpu", enabled=False):
        if all(t.dtype == torch.float32 for t in (w1, b1, w2, b2)):        # autocast over fp32 parameters: through their per-step shadows
            return FusedMlpFunc.apply(x.to(dt).contiguous(), w1, b1, w2, b2, dt)
        return FusedMlpFunc.apply(x.to(dt).contiguous(), w1.to(dt).contiguous(), b1.to(dt), w2.to(dt).contiguous(), b2.to(dt))


def _refuse(name):
    class _Missing(nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError(f"flash_attn.{name} (tensor / sequence parallel or fused-dense variants) is not part of the "
                                      "HyenaDNA path: every hg38 configuration builds the plain module (process_group=None)")
    _Missing.__name__ = name.rsplit(".", 1)[-1]
    return _Missing


FusedMLP = _refuse("modules.mlp.FusedMLP")
ParallelFusedMLP = _refuse("modules.mlp.ParallelFusedMLP")
ParallelMHA = _refuse("modules.mha.ParallelMHA")
ParallelGPT2Embeddings = _refuse("modules.embedding.ParallelGPT2Embeddings")


# ---------------------------------------------------------------------------------------------------------------------
# flash_attn.modules.mha.MHA  (semantics: simple_lm.py:26-150; HyenaDNA sets attn_layer_idx = None, so this is only
# ever built by hybrid configurations)
# ---------------------------------------------------------------------------------------------------------------------
class MHA(nn.Module):
    def __init__(self, embed_dim, num_heads, bias=True, dropout=0.0, softmax_scale=None, causal=False, layer_idx=None,
                 dwconv=False, return_residual=False, device=None, dtype=None, **unused):
        factory_kwargs = {"device": device, "dtype": dtype}
        super().__init__()
        assert embed_dim % num_heads == 0, "self.kdim must be divisible by num_heads"
        self.embed_dim, self.num_heads, self.head_dim = embed_dim, num_heads, embed_dim // num_heads
        self.causal, self.layer_idx, self.dwconv, self.return_residual = causal, layer_idx, dwconv, return_residual
        self.softmax_scale, self.dropout_p = softmax_scale, dropout
        self.Wqkv = nn.Linear(embed_dim, 3 * embed_dim, bias=bias, **factory_kwargs)
        if dwconv:
            self.dwconv_qkv = nn.Conv1d(3 * embed_dim, 3 * embed_dim, kernel_size=3, padding=2, groups=3 * embed_dim)
        self.out_proj = nn.Linear(embed_dim, embed_dim, **factory_kwargs)

    def forward(self, x, key_padding_mask=None, **kwargs):
        qkv = self.Wqkv(x)
        if self.dwconv:
            qkv = self.dwconv_qkv(qkv.transpose(1, 2))[..., :-2].transpose(1, 2).contiguous()
        B, S, _ = qkv.shape
        q, k, v = qkv.view(B, S, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)      # (3, B, H, S, d)
        mask = None
        if key_padding_mask is not None:
            mask = key_padding_mask[:, None, None, :]
            if self.causal:
                mask = mask & torch.ones(S, S, dtype=torch.bool, device=x.device).tril()
        ctx = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=self.dropout_p if self.training else 0.0,
                                             is_causal=self.causal and mask is None, scale=self.softmax_scale)
        out = self.out_proj(ctx.transpose(1, 2).reshape(B, S, self.embed_dim))
        return out if not self.return_residual else (out, x)


# ---------------------------------------------------------------------------------------------------------------------
# flash_attn.modules.embedding.GPT2Embeddings  (semantics: simple_lm.py:153-190)
# ---------------------------------------------------------------------------------------------------------------------
class SmallVocabEmbeddingFunc(torch.autograd.Function):
    """``F.embedding`` whose weight gradient is a dense product ``onehot(ids)^T @ dh``.  PyTorch's embedding backward sorts /
    scatters per token; with the 16-row DNA vocabulary and 10^6 tokens per sequence every token collides with 2.6e5 others
    and that kernel alone took 10.5 ms of a 212 ms model step (profiles/r2h).  The one-hot product is one slice-batched
    GEMM (``projection.split_k_weight_grad``), deterministic."""

    @staticmethod
    def forward(ctx, ids, weight):
        ctx.save_for_backward(ids)
        ctx.vocab = weight.shape[0]
        return F.embedding(ids, weight)

    @staticmethod
    def backward(ctx, dh):
        from .projection import split_k_weight_grad
        (ids,) = ctx.saved_tensors
        dh2 = dh.reshape(-1, dh.shape[-1])
        onehot = F.one_hot(ids.reshape(-1), ctx.vocab).to(dh2.dtype)
        return None, split_k_weight_grad(onehot, dh2).to(dh.dtype)


def small_vocab_embedding(ids, emb):
    if (ids.is_cuda and emb.padding_idx is None and emb.max_norm is None and not emb.sparse and emb.num_embeddings <= 64
            and ids.numel() >= 32768):
        return SmallVocabEmbeddingFunc.apply(ids, emb.weight)
    return emb(ids)


class GPT2Embeddings(nn.Module):
    def __init__(self, embed_dim, vocab_size, max_position_embeddings, padding_idx=None, word_embed_proj_dim=None,
                 device=None, dtype=None):
        factory_kwargs = {"device": device, "dtype": dtype}
        super().__init__()
        if word_embed_proj_dim is None:
            self.word_embeddings = nn.Embedding(vocab_size, embed_dim, padding_idx=padding_idx, **factory_kwargs)
            self.project_in = None
        else:
            self.word_embeddings = nn.Embedding(vocab_size, word_embed_proj_dim, padding_idx=padding_idx, **factory_kwargs)
            self.project_in = nn.Linear(word_embed_proj_dim, embed_dim, bias=False, **factory_kwargs)
        self.max_position_embeddings = max_position_embeddings
        if self.max_position_embeddings > 0:
            self.position_embeddings = nn.Embedding(max_position_embeddings, embed_dim, **factory_kwargs)

    def forward(self, input_ids, position_ids=None):
        batch_size, seqlen = input_ids.shape
        embeddings = small_vocab_embedding(input_ids, self.word_embeddings)
        if self.project_in is not None:
            embeddings = self.project_in(embeddings)
        if self.max_position_embeddings > 0:
            if position_ids is None:
                position_ids = torch.arange(seqlen, dtype=torch.long, device=input_ids.device)
            embeddings = embeddings + self.position_embeddings(position_ids)
        return embeddings


# ---------------------------------------------------------------------------------------------------------------------
# flash_attn.modules.block.Block  (semantics: simple_lm.py:214-305; constructor as long_conv_lm.py:171-185 calls it)
# ---------------------------------------------------------------------------------------------------------------------
class Block(nn.Module):
    def __init__(self, dim, mixer_cls=None, mlp_cls=None, norm_cls=nn.LayerNorm, dropout_cls=nn.Dropout, prenorm=True,
                 resid_dropout1=0.0, resid_dropout2=0.0, drop_path1=0.0, drop_path2=0.0, fused_dropout_add_ln=False,
                 return_residual=False, residual_in_fp32=False, sequence_parallel=False, mark_shared_params=False):
        super().__init__()
        if drop_path1 or drop_path2:
            raise NotImplementedError("stochastic depth is not used by any HyenaDNA configuration")
        if sequence_parallel or mark_shared_params:
            raise NotImplementedError("tensor / sequence parallelism is not part of the HyenaDNA path (north_star: data parallel only)")
        self.prenorm = prenorm
        self.fused_dropout_add_ln = fused_dropout_add_ln
        self.return_residual = return_residual
        self.residual_in_fp32 = residual_in_fp32
        if self.residual_in_fp32:
            assert self.prenorm, "residual_in_fp32 is only compatible with prenorm=True"
        if mixer_cls is None:
            mixer_cls = partial(MHA, num_heads=dim // 64)
        if mlp_cls is None:
            mlp_cls = partial(Mlp, hidden_features=4 * dim)
        self.mixer = mixer_cls(dim)
        self.dropout1 = dropout_cls(resid_dropout1)
        self.norm1 = norm_cls(dim)
        self.mlp = mlp_cls(dim)
        if not isinstance(self.mlp, nn.Identity):
            self.dropout2 = dropout_cls(resid_dropout2)
            self.norm2 = norm_cls(dim)

    def _add_norm(self, hidden_states, residual, drop, norm):
        if self.fused_dropout_add_ln:
            return dropout_add_layer_norm(hidden_states, residual, norm.weight, norm.bias, drop.p if self.training else 0.0,
                                          norm.eps, prenorm=True, residual_in_fp32=self.residual_in_fp32)
        dropped = drop(hidden_states)
        residual = (dropped + residual) if residual is not None else dropped
        hidden_states = norm(residual.to(dtype=norm.weight.dtype))
        if self.residual_in_fp32:
            residual = residual.to(torch.float32)
        return hidden_states, residual

    def forward(self, hidden_states, residual=None, mixer_subset=None, mixer_kwargs=None, normed=False):
        """``normed``: (hidden_states, residual) are already the outputs of this block's first dropout -> add -> LayerNorm (HyenaDNALM hands
        them over when that pass also gathered the token embedding)"""
        if self.prenorm:
            if not normed:
                hidden_states, residual = self._add_norm(hidden_states, residual, self.dropout1, self.norm1)
            mixer_kwargs = {} if mixer_kwargs is None else mixer_kwargs
            if mixer_subset is not None:
                mixer_kwargs["mixer_subset"] = mixer_subset
            fused = None
            if (not mixer_kwargs and not isinstance(self.mlp, nn.Identity) and self.fused_dropout_add_ln and self.residual_in_fp32
                    and (self.dropout2.p == 0.0 or not self.training) and hasattr(self.mixer, "forward_add_norm")
                    and isinstance(self.norm2, nn.LayerNorm) and self.norm2.weight is not None):
                # the mixer's last kernel carries this block's second add + LayerNorm in its epilogue (hyena.HyenaOperator.forward_add_norm)
                fused = self.mixer.forward_add_norm(hidden_states, residual, self.norm2.weight, self.norm2.bias, self.norm2.eps)
            if fused is not None:
                hidden_states, residual = fused
                return self.mlp(hidden_states), residual
            hidden_states = self.mixer(hidden_states, **mixer_kwargs)
            if mixer_subset is not None:
                residual = residual[:, mixer_subset]
            if not isinstance(self.mlp, nn.Identity):
                hidden_states, residual = self._add_norm(hidden_states, residual, self.dropout2, self.norm2)
                hidden_states = self.mlp(hidden_states)
            return hidden_states, residual
        assert residual is None
        mixer_out = self.mixer(hidden_states, **(mixer_kwargs if mixer_kwargs is not None else {}))
        if self.return_residual:
            mixer_out, hidden_states = mixer_out
        hidden_states = self.norm1((self.dropout1(mixer_out) + hidden_states).to(dtype=self.norm1.weight.dtype))
        if not isinstance(self.mlp, nn.Identity):
            mlp_out = self.mlp(hidden_states)
            if self.return_residual:
                mlp_out, hidden_states = mlp_out
            hidden_states = self.norm2((self.dropout2(mlp_out) + hidden_states).to(dtype=self.norm2.weight.dtype))
        return hidden_states


# ---------------------------------------------------------------------------------------------------------------------
# flash_attn.utils.generation.GenerationMixin / flash_attn.utils.distributed
# ---------------------------------------------------------------------------------------------------------------------
class GenerationMixin:
    """Greedy / top-k sampling by re-running the causal model on the growing prefix (the convolutional mixer has no
    incremental state in the reference either: HyenaOperator ignores ``inference_params``)."""

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        return None

    @torch.no_grad()
    def generate(self, input_ids, max_length, top_k=1, temperature=1.0, return_dict_in_generate=False, output_scores=False, **kwargs):
        ids, scores = input_ids, []
        while ids.shape[1] < max_length:
            out = self(ids)
            logits = (out[0] if isinstance(out, tuple) else out)
            logits = (logits.logits if hasattr(logits, "logits") else logits)[:, -1] / max(temperature, 1e-6)
            if top_k <= 1:
                nxt = logits.argmax(-1, keepdim=True)
            else:
                v, i = logits.topk(top_k, dim=-1)
                nxt = i.gather(-1, torch.multinomial(torch.softmax(v.float(), -1), 1))
            scores.append(logits)
            ids = torch.cat([ids, nxt], dim=1)
        if return_dict_in_generate:
            return namedtuple("GreedySearchDecoderOnlyOutput", ["sequences", "scores"])(ids, tuple(scores) if output_scores else None)
        return ids


def sync_shared_params(model, process_group):
    """Broadcast parameters marked ``_shared_params`` from rank 0 of the group (flash_attn.utils.distributed)."""
    import torch.distributed as dist
    for _, p in sorted(model.named_parameters()):
        if getattr(p, "_shared_params", False):
            with torch.no_grad():
                dist.broadcast(p, src=dist.get_global_rank(process_group, 0), group=process_group)


def all_gather_raw(input_, process_group, async_op=False):
    import torch.distributed as dist
    world = dist.get_world_size(process_group)
    out = torch.empty(world * input_.shape[0], *input_.shape[1:], dtype=input_.dtype, device=input_.device)
    handle = dist.all_gather_into_tensor(out, input_.contiguous(), group=process_group, async_op=async_op)
    return out, handle


# ---------------------------------------------------------------------------------------------------------------------
# The hyenadna-* language model without Lightning / Hydra (structure: long_conv_lm.py:249-502, hg38_hyena.yaml)
# ---------------------------------------------------------------------------------------------------------------------
def _init_weights(module, n_layer, initializer_range=0.02, rescale_prenorm_residual=True):
    """long_conv_lm.py:204-246 (GPT-2 scheme: out_proj / fc2 scaled by 1 / sqrt(2 n_layer))."""
    if isinstance(module, nn.Linear):
        nn.init.normal_(module.weight, std=initializer_range)
        if module.bias is not None:
            nn.init.zeros_(module.bias)
    elif isinstance(module, nn.Embedding):
        nn.init.normal_(module.weight, std=initializer_range)
    if rescale_prenorm_residual:
        for name, p in module.named_parameters():
            if name in ("out_proj.weight", "fc2.weight"):
                nn.init.normal_(p, mean=0.0, std=initializer_range / math.sqrt(2 * n_layer))


class HyenaDNALM(nn.Module, GenerationMixin):
    """``ConvLMHeadModel`` with the Hyena mixer in every block (the hg38 pre-training model): same sub-module names as the
    reference (``backbone.embeddings.word_embeddings``, ``backbone.layers.N.{mixer,norm1,norm2,mlp.fc1,mlp.fc2}``,
    ``backbone.ln_f``, ``lm_head``), so a reference checkpoint's state dict loads."""

    def __init__(self, d_model, n_layer, d_inner, vocab_size, layer=None, max_position_embeddings=0, resid_dropout=0.0,
                 embed_dropout=0.1, layer_norm_epsilon=1e-5, initializer_cfg=None, fused_dropout_add_ln=True,
                 residual_in_fp32=True, pad_vocab_size_multiple=1, checkpoint_mixer=False, checkpoint_mlp=False, **unused):
        super().__init__()
        from .hyena import HyenaOperator
        # Keywords of ConvLMHeadModel / LMBackbone (long_conv_lm.py:249-272, 402-424) this single-process, Hyena-only model has
        # no counterpart for are accepted at their "off" values only: a config asking for attention layers, tensor parallelism
        # or flash_attn's FusedMLP must not silently get a different model.
        off = {"attn_layer_idx": None, "attn_cfg": None, "process_group": None, "fused_mlp": False, "identity_mlp": False,
               "sequence_parallel": (True, False), "return_hidden_state": False}
        # the factory keywords every reference module takes (long_conv_lm.py:266: factory_kwargs): applied after construction
        device, dtype = unused.pop("device", None), unused.pop("dtype", None)
        for key, val in unused.items():
            allowed = off.get(key, KeyError)
            if allowed is KeyError:
                raise TypeError(f"HyenaDNALM: unknown keyword {key!r}")
            if not (val in allowed if isinstance(allowed, tuple) else val == allowed):
                raise NotImplementedError(f"HyenaDNALM: {key}={val!r} is not supported (Hyena mixers, one process per GPU, "
                                          "library-GEMM MLP only)")
        if vocab_size % pad_vocab_size_multiple != 0:
            vocab_size += pad_vocab_size_multiple - (vocab_size % pad_vocab_size_multiple)
        layer = dict(layer or {})
        layer.pop("_name_", None)
        self.d_model, self.residual_in_fp32, self.fused_dropout_add_ln = d_model, residual_in_fp32, fused_dropout_add_ln
        backbone = nn.Module()
        backbone.embeddings = GPT2Embeddings(d_model, vocab_size, max_position_embeddings)
        mlp_cls = partial(Mlp, hidden_features=d_inner if d_inner is not None else 4 * d_model,
                          activation=partial(F.gelu, approximate="tanh"))
        norm_cls = partial(nn.LayerNorm, eps=layer_norm_epsilon)
        backbone.layers = nn.ModuleList([
            Block(d_model, partial(HyenaOperator, **layer), mlp_cls, norm_cls=norm_cls, prenorm=True,
                  resid_dropout1=embed_dropout if i == 0 else resid_dropout, resid_dropout2=resid_dropout,
                  fused_dropout_add_ln=fused_dropout_add_ln, residual_in_fp32=residual_in_fp32) for i in range(n_layer)])
        # long_conv_lm.py:196-199: activation checkpointing wraps the sub-module, which also moves its state-dict keys under
        # `.layer` -- a checkpoint saved with these flags loads only into a model built with them
        for blk in backbone.layers:
            if checkpoint_mlp:
                blk.mlp = CheckpointedModule(blk.mlp)
            if checkpoint_mixer:
                blk.mixer = CheckpointedModule(blk.mixer)
        backbone.drop_f = nn.Dropout(resid_dropout)
        backbone.ln_f = nn.LayerNorm(d_model, eps=layer_norm_epsilon)
        self.backbone = backbone
        self.checkpoint_mixer, self.checkpoint_mlp = checkpoint_mixer, checkpoint_mlp
        self.lm_head = nn.Linear(d_model, vocab_size, bias=False)
        self.apply(partial(_init_weights, n_layer=n_layer, **(initializer_cfg or {})))
        self.tie_weights()
        if device is not None or dtype is not None:
            self.to(device=device, dtype=dtype)

    def tie_weights(self):
        self.lm_head.weight = self.backbone.embeddings.word_embeddings.weight

    def hidden(self, input_ids, position_ids=None):
        bb = self.backbone
        emb, blk0 = bb.embeddings, bb.layers[0]
        if (self.fused_dropout_add_ln and self.residual_in_fp32 and isinstance(blk0, Block) and blk0.prenorm and blk0.fused_dropout_add_ln
                and blk0.residual_in_fp32 and emb.project_in is None and emb.max_position_embeddings <= 0
                and embedding_fusable(input_ids, emb.word_embeddings, blk0.norm1.weight)):
            # the token embedding gathered inside the first block's dropout -> add -> LayerNorm pass (block.py): neither the (B, L, D) fp32
            # embedding nor its gradient exists; the normed output leaves in the autocast type the mixer would round it to anyway
            dev_type = input_ids.device.type
            odt = torch.get_autocast_dtype(dev_type) if torch.is_autocast_enabled(dev_type) else torch.float32
            if odt not in (torch.bfloat16, torch.float16):
                odt = torch.float32
            hidden_states, residual = embedding_dropout_add_layer_norm(input_ids, emb.word_embeddings.weight, blk0.norm1.weight, blk0.norm1.bias,
                                                                       blk0.dropout1.p if self.training else 0.0, blk0.norm1.eps, out_dtype=odt)
            hidden_states, residual = blk0(hidden_states, residual, normed=True)
            rest = list(bb.layers)[1:]
        else:
            hidden_states, residual = emb(input_ids, position_ids=position_ids), None
            rest = bb.layers
        for blk in rest:
            hidden_states, residual = blk(hidden_states, residual)
        if self.fused_dropout_add_ln:
            return dropout_add_layer_norm(hidden_states, residual, bb.ln_f.weight, bb.ln_f.bias,
                                          bb.drop_f.p if self.training else 0.0, bb.ln_f.eps, prenorm=False,
                                          residual_in_fp32=self.residual_in_fp32)
        dropped = bb.drop_f(hidden_states)
        residual = (dropped + residual) if residual is not None else dropped
        return bb.ln_f(residual.to(dtype=bb.ln_f.weight.dtype))

    def _aligned_length(self, input_ids):
        """Round 6: several sequences of a length that is not a multiple of 64 -- the reference trainer's own batches: L = max_length - 1
        (hg38_dataset.py:222), B = 256 / 8 / 2 (hg38_hyena.yaml:47-48) -- run on sequences padded at the END to the next multiple of 64.  Every
        operation of the model is causal or per-position (embedding, LayerNorm, MLP, short conv, long conv, gates), so positions < L see exactly the
        values of the unpadded run (the long convolution may pick a different transform size: fp32 rounding level); the pad positions' logits
        are dropped before anybody sees them, so they receive zero gradients and contribute nothing to any weight gradient.  What it buys: inside a
        channel row of the flattened (C, B L) layout the rows of odd-length sequences start 2 bytes off every 4 / 16-byte boundary -- one layer at
        32767 x 8 ran 11 % slower than at 32768 x 8, the step at 1023 x 256 x 128 6 % slower (profiles/r6a_bench_default.json).  Only when every
        layer's l_max admits the padded length (hg38 configurations: l_max = max_length + 2) and the fused kernels are in use."""
        B, L = input_ids.shape
        # ONE odd-length sequence: its rows are aligned (pitched), what is left are the library weight-gradient products over an odd number of
        # positions (2^20 - 1: dW1 712 vs 612 us, dW_in 611 vs 571, dW_out 303 vs 260 per layer: profiles/r6_wgrad_plan.txt).  Padded where that
        # turns them into the aligned one-level plan -- the padded length a multiple of 4096 (2^20 - 1, 32767: model step 157.5 -> 156.3 ms at
        # 2^20 - 1); at 999 999 / 449 999 the padded count still needs the two-level plan and padding bought nothing (profiles/r6aa_*)
        if not PAD_SEQUENCES or L % _SEQ_ALIGN == 0 or L < _SEQ_ALIGN:
            return L
        if B <= 1 and (L < _PAD_SINGLE_MIN or (L + (-L) % _SEQ_ALIGN) % _PAD_SINGLE_ROWS != 0):
            return L
        from . import _lib
        if not (input_ids.is_cuda or _lib._backend.name != "hip"):
            return L
        Lp = L + (-L) % _SEQ_ALIGN
        emb = self.backbone.embeddings
        if emb.max_position_embeddings > 0 and Lp > emb.max_position_embeddings:
            return L
        for blk in self.backbone.layers:
            mixer = getattr(blk.mixer, "layer", blk.mixer)
            if getattr(mixer, "l_max", Lp) < Lp:
                return L
        return Lp

    def forward(self, input_ids, position_ids=None, inference_params=None, state=None):
        # (the head through projection.hyena_linear: its weight gradient contracts 16 x 256 outputs over 10^6 tokens, which the GEMM library
        # runs on 16 workgroups -- 1.19 ms per step at 2^20 tokens, profiles/r4y_model_stats.csv -- and the split-K form does not)
        L = input_ids.shape[1]
        Lp = self._aligned_length(input_ids) if position_ids is None else L
        if Lp != L:
            input_ids = F.pad(input_ids, (0, Lp - L), value=0)     # any valid token id: the pad positions' outputs are dropped
        lm_logits = hyena_linear(self.hidden(input_ids, position_ids), self.lm_head.weight, self.lm_head.bias)
        if Lp != L:
            lm_logits = lm_logits[:, :L]                 # (a view: the pad positions' logits reach nobody -> zero gradients)
        return namedtuple("CausalLMOutput", ["logits"])(logits=lm_logits), None

    def loss(self, input_ids, targets, ignore_index=-100):
        """next-token cross entropy (src/tasks/metrics.py cross_entropy over the flattened logits), logits in fp32"""
        logits = self.forward(input_ids)[0].logits
        return token_cross_entropy(logits, targets, ignore_index=ignore_index)


def token_cross_entropy(logits, targets, ignore_index=-100):
    """``F.cross_entropy(logits.float().reshape(-1, V), targets.reshape(-1), ignore_index=ignore_index)`` -- the mean of -log p[target] over
    the tokens that are not ignored -- written as log-softmax + gather + masked mean.  Same value and gradients to fp32 summation order; on
    ROCm PyTorch's own reduction for this 2-D case (nll_loss_forward_reduce_cuda_kernel_2d) runs in ONE workgroup: 1.07 ms forward and 0.8 ms
    backward per step at 2^20 tokens x 16 classes (profiles/r4y_model_stats.csv) for 64 MB of logits; these are streaming kernels."""
    lp = torch.log_softmax(logits.float().reshape(-1, logits.shape[-1]), dim=-1)
    t = targets.reshape(-1)
    valid = t != ignore_index
    picked = lp.gather(1, t.clamp_min(0).unsqueeze(1)).squeeze(1)
    return -(picked * valid).sum() / valid.sum()


class CheckpointedModule(nn.Module):
    """long_conv_lm.py:39-45: the wrapped module under the attribute `layer`, its forward recomputed in the backward."""

    def __init__(self, layer):
        super().__init__()
        self.layer = layer

    def forward(self, x):
        from torch.utils.checkpoint import checkpoint
        return checkpoint(self.layer, x, use_reentrant=False)


class GraphedTrainStep:
    """One whole training step -- forward, loss, backward, optimizer step -- captured into ONE hipGraph and replayed.

    Below L ~ 32k the eager step of the 8-layer model is bound by the ~1 500 kernel launches Python issues, not by the GPU
    (DESIGN.md section 5); a replay costs one launch.  Everything on this path is stream-ordered and allocation-free after the
    warm-up (include/hyena_fftconv.h), so the capture needs nothing special: static input buffers, gradients that live inside
    the graph's memory pool, and an optimizer built with ``capturable=True`` (its step counters are device tensors).

        opt  = torch.optim.AdamW(model.parameters(), lr=6e-4, weight_decay=0.1, capturable=True)
        step = GraphedTrainStep(model, opt, input_ids, targets)      # warms up (3 eager steps) and captures
        loss = step(next_ids, next_targets)                           # copies into the static buffers, replays

    The captured step always sees batches of the captured shape; to change the learning rate between replays make it a
    device tensor (``lr=torch.tensor(6e-4, device=...)``) and update it in place.  Drop the loss / outputs of earlier eager
    steps before constructing this (their autograd graphs pin gradient accumulators to another stream), and see
    ``hyena_dna_amd/__init__.py`` for the ROCm runtime setting whole-step replays need (applied at import).
    """

    def __init__(self, model, optimizer, input_ids, targets, autocast_dtype=torch.bfloat16, warmup=3, ignore_index=-100,
                 clip_grad_norm=0.0):
        if not input_ids.is_cuda:
            raise RuntimeError("GraphedTrainStep captures a hipGraph: model and batch must live on a ROCm device")
        for group in optimizer.param_groups:
            if not group.get("capturable", False):
                raise RuntimeError("GraphedTrainStep needs an optimizer built with capturable=True")
        import hyena_dna_amd
        if not hyena_dna_amd.GRAPH_SAFE or os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") != "0":
            raise RuntimeError("GraphedTrainStep: hipGraph replays of a whole step are only reliable on this ROCm runtime with "
                               "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in the environment before the HIP runtime initialises "
                               "(call hyena_dna_amd.prepare_graph_runtime() before the first torch.cuda call, or export the "
                               "variable; a process that exported it at start-up but touched torch.cuda before importing this package says so "
                               "with HYENA_GRAPH_SAFE_OVERRIDE=1); see hyena_dna_amd/__init__.py and INTEGRATION.md section 6")
        self.model, self.optimizer = model, optimizer
        self.ids, self.targets = input_ids.clone(), targets.clone()
        self.autocast_dtype, self.ignore_index = autocast_dtype, ignore_index
        self.clip_grad_norm = float(clip_grad_norm or 0.0)           # trainer.gradient_clip_val (hg38_hyena.yaml:41), inside the graph
        # Gradient-accumulation nodes remember the stream they were created on.  Nodes left over from eager steps on another
        # stream (kept alive by a retained loss tensor, or by not-yet-collected reference cycles of autograd contexts) would run
        # OUTSIDE the capture: drop them, and warm up on the very stream the capture uses.
        import gc
        gc.collect()
        from . import _lib
        # keep-the-spectra decisions are re-taken during this warm-up, with the model (and, from its second step on, the optimizer state)
        # already allocated -- not inherited from whatever memory was free when a shape was first seen; they then stay fixed for the capture
        _lib.reset_save_decisions(input_ids.device)
        side = torch.cuda.Stream(input_ids.device)
        self._side = side
        side.wait_stream(torch.cuda.current_stream(input_ids.device))
        with torch.cuda.stream(side):                 # twiddle tables, workspaces, optimizer state, GEMM heuristics: all created here
            # The warm-up runs REAL optimizer updates (the optimizer's state tensors must exist, on this stream, before the capture).  They
            # are not training steps: parameters, Adam moments and step counters are put back afterwards, in place (the graph will hold
            # these very tensors), so that a graphed run and an eager run of the same config + seed start from the same state.
            params = [p for g in optimizer.param_groups for p in g["params"]]
            with torch.no_grad():
                snap_p = [p.detach().clone() for p in params]
            had_state = {id(p): {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in optimizer.state.get(p, {}).items()}
                         for p in params}
            for _ in range(max(1, int(warmup))):
                optimizer.zero_grad(set_to_none=True)
                self._fwd_bwd()
                optimizer.step()
            optimizer.zero_grad(set_to_none=True)      # the gradients of the captured step live in the graph's pool
            with torch.no_grad():
                for p, q in zip(params, snap_p):
                    p.copy_(q)
                for p in params:
                    st, old = optimizer.state.get(p, {}), had_state[id(p)]
                    for k, v in st.items():
                        if torch.is_tensor(v):
                            if k in old and torch.is_tensor(old[k]):
                                v.copy_(old[k])
                            else:
                                v.zero_()                  # fresh optimizer: moments and the step counter start at zero
            del snap_p, had_state
            gc.collect()
            side.synchronize()
            from . import _castcache
            _castcache.invalidate()                    # the batched weight cast becomes part of the captured step
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=side):
                self.loss = self._fwd_bwd().detach()
                optimizer.step()
        torch.cuda.current_stream(input_ids.device).wait_stream(side)

    def release(self):
        """Drop the graph and what the long-convolution binding keeps for its capture stream (workspace, tables stay: they are per length).
        Call before building the next GraphedTrainStep of a sequence-length stage."""
        from . import _lib
        dev = self.ids.device
        self.graph = None
        self.loss = None
        torch.cuda.synchronize(dev)
        _lib.release_stream_state(dev, self._side.cuda_stream)
        self._side = None

    def _fwd_bwd(self):
        enabled = self.autocast_dtype is not None and self.autocast_dtype != torch.float32
        with torch.autocast("cuda", dtype=self.autocast_dtype if enabled else torch.bfloat16, enabled=enabled):
            loss = self.model.loss(self.ids, self.targets, ignore_index=self.ignore_index)
        loss.backward()
        if self.clip_grad_norm > 0:                                   # device-side norm and scale: no host sync, capturable
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.clip_grad_norm)
        return loss

    def __call__(self, input_ids=None, targets=None):
        """Replays the step on (input_ids, targets) -- or on the batch already in the static buffers -- and returns the
        (static, device-resident) loss tensor; read it with .item() only when you need the number."""
        if input_ids is not None:
            self.ids.copy_(input_ids, non_blocking=True)
        if targets is not None:
            self.targets.copy_(targets, non_blocking=True)
        self.graph.replay()
        return self.loss
