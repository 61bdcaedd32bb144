"""Module seam of the hot path: drop-in counterparts of the reference's ``HyenaOperator`` / ``HyenaFilter``
(``src/models/sequence/hyena.py:158-448``; standalone copy ``standalone_hyenadna.py:147-293``).

Same constructor keywords (stray ones such as ``layer_idx``/``device``/``dtype`` are tolerated exactly as the
reference tolerates them via ``**filter_args`` -> ``**kwargs``, hyena.py:290,373-380,177,221,143), same parameter
and buffer names (so reference checkpoints load with ``load_state_dict``), same ``_optim`` per-parameter
hyper-parameter tags (src/utils/train.py:142-156), same forward semantics ``(B, L, D) -> (B, L', D)``.  The long
convolution is ``hyena_dna_amd.fftconv.fftconv_func`` (HIP); there is no torch.fft path here.

Register with the reference's name-based instantiate (src/utils/registry.py:40-41) by
``registry.layer["hyena"] = "hyena_dna_amd.hyena.HyenaOperator"``.
"""
import math
import os

import torch
import torch.nn as nn

from .fftconv import fftconv_func, fftconv_ref
from .filter import fused_filter_ok, hyena_filter_dl
from .mixer import hyena_mixer_core, hyena_mixer_core_cm, hyena_mixer_core_cm_order_n, hyena_mixer_out_cm, mixer_out_supported
from .projection import hyena_linear, in_proj_cm, in_proj_pre_cm, out_proj_cm

# Layout of the tensors between the operator's two projections: channel-major (x^T written by the in_proj GEMM, z^T read by
# the out_proj GEMM, no transposes anywhere: csrc/cm_kernels.h) or the reference's position-major (B, L, 3D) with the
# transposes fused into the shell kernels (csrc/mixer_kernels.h).  HYENA_MIXER_LAYOUT=position selects the latter (A/B).
CHANNEL_MAJOR = os.environ.get("HYENA_MIXER_LAYOUT", "channel").lower() != "position"
# out_proj's kernel can carry the block's residual add + LayerNorm in its epilogue (HyenaOperator.forward_add_norm; bit-identical results).  Round 5's
# kernel measured slower everywhere (profiles/r5c_outproj_addnorm_not_kept.txt); round 6's generation 2 (csrc/proj2_kernels.h: whole rows per wavefront,
# residual rows prefetched by LDS-direct loads) wins at d_model 128 with many short sequences -- the shipped experiment's shape: 1024 x 256 x 128: 110 vs 131 us
# for the two launches, 1023 x 256: 121 vs 169 -- and still loses at d_model 256 (2^20: 1266 - 1270 vs 1120 - 1144 us; 32768 x 8: 298 vs 255;
# profiles/r6l_bench_outproj_ln_lds_residual.txt): "auto" (default) = fused at d_model 128 only; HYENA_ADD_NORM_FUSED=1 / 0 forces it.
ADD_NORM_FUSED = {"1": True, "0": False}.get(os.environ.get("HYENA_ADD_NORM_FUSED", "auto"), "auto")
# order >= 3 (configs/model/layer/hyena_dna.yaml:3): the channel-major route of mixer.HyenaMixerCMOrderNFunc -- both projections as channel-major GEMMs,
# every gate one of the order-2 shell kernels on a row view of x^T, nothing transposed -- instead of the reference's op-by-op graph around the HIP
# convolution (profiles/r6t_order3.txt).  HYENA_ORDER_N_FUSED=0: the generic route (A/B).
ORDER_N_FUSED = os.environ.get("HYENA_ORDER_N_FUSED", "1") != "0"


# The implicit filter depends on parameters only.  Its kernels are latency-bound (2 wavefronts per SIMD, eight dependent round trips per tile: 2.5 - 2.9 TB/s),
# the projections next to it are bandwidth-bound: on a SECOND STREAM the filter's forward overlaps in_proj, and -- autograd runs a node's backward on the
# stream of its forward -- its backward overlaps the projections' input / weight gradient GEMMs (round 6: model step 153.5 -> 151.0 ms at 2^20 x 256,
# 34.1 -> 33.5 at 32768 x 8, 48.5 -> 47.2 at 159999 x 2; profiles/r6af_filter_side_stream.txt).  Results are bit-identical.  "auto" (default): on for
# device tensors of at least _FILTER_SIDE_MIN_L positions outside a stream capture and outside multi-process jobs (DistributedDataParallel launches
# its bucket all-reduces from whichever stream marks a bucket ready: gradients written on a second stream are not this package's to order there);
# HYENA_FILTER_SIDE_STREAM=1 / 0 forces it on (still never inside a capture) / off.
FILTER_SIDE_STREAM = {"1": True, "0": False}.get(os.environ.get("HYENA_FILTER_SIDE_STREAM", "auto"), "auto")
_FILTER_SIDE_MIN_L = 8192
_side_streams = {}
_device_bytes = {}


def _filter_side_stream(u, l):
    if FILTER_SIDE_STREAM is False or not (u.is_cuda and CHANNEL_MAJOR) or torch.cuda.is_current_stream_capturing():
        return None
    if FILTER_SIDE_STREAM == "auto":
        if l < _FILTER_SIDE_MIN_L:
            return None
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return None
        # the second stream has an allocator pool of its own (2^20 x 256, 8 layers: 207 GiB reserved instead of 133): not when memory is already tight
        total = _device_bytes.get(u.device.index)
        if total is None:
            total = _device_bytes[u.device.index] = torch.cuda.get_device_properties(u.device).total_memory
        if torch.cuda.memory_reserved(u.device) > 0.8 * total:
            return None
    s = _side_streams.get(u.device.index)
    if s is None:
        s = _side_streams[u.device.index] = torch.cuda.Stream(u.device)
    return s


class _FilterOnSideStream:
    """f = _FilterOnSideStream(u, l);  k = f.run(lambda: filter_dl(...));  ... in_proj on the caller's stream ...;  f.join(k)
    run: the second stream first waits for the caller's (the parameters' last update happened there), then evaluates the filter -- PyTorch records the
    autograd node's stream, so its backward runs there too, after the engine has made it wait for the gradient's producer.  join: the caller's stream
    waits for the filter and the result's memory is marked as used on it (the allocator must not recycle it while the convolution reads it)."""

    def __init__(self, u, l):
        self.side = _filter_side_stream(u, l)
        self.cur = torch.cuda.current_stream(u.device) if self.side is not None else None

    def run(self, fn):
        if self.side is None:
            return fn()
        self.side.wait_stream(self.cur)                   # (the parameters' last update happened on the caller's stream)
        with torch.cuda.stream(self.side):
            return fn()

    def join(self, ks):
        if self.side is not None:
            self.cur.wait_stream(self.side)
            for k in (ks if isinstance(ks, (list, tuple)) else [ks]):
                k.record_stream(self.cur)
        return ks


def _add_norm_fused(d_model):
    if ADD_NORM_FUSED == "auto":
        from . import _lib
        return d_model == 128 and _lib.proj_kernel_generation(0) == 2
    return bool(ADD_NORM_FUSED)


__all__ = ["HyenaOperator", "HyenaFilter", "PositionalEmbedding", "ExponentialModulation", "Sin"]


def _fused_filter_requires_gpu():
    from . import _lib
    return _lib._backend.name == "hip"


class _OptimModule(nn.Module):
    """register(name, tensor, lr, wd): buffer when lr == 0, else a Parameter tagged with ``_optim``
    (reference: src/utils/train.py:142-156; consumed by train.py:443-468)."""

    def register(self, name, tensor, lr=None, wd=0.0):
        if lr == 0.0:
            self.register_buffer(name, tensor)
            return
        self.register_parameter(name, nn.Parameter(tensor))
        tag = {}
        if lr is not None:
            tag["lr"] = lr
        if wd is not None:
            tag["weight_decay"] = wd
        getattr(self, name)._optim = tag


class Sin(nn.Module):
    """sin(freq * x) with a (1, dim) frequency, trainable by default (hyena.py:96-106)."""

    def __init__(self, dim, w=10, train_freq=True):
        super().__init__()
        init = w * torch.ones(1, dim)
        self.freq = nn.Parameter(init) if train_freq else init

    def forward(self, x):
        return torch.sin(self.freq * x)


class PositionalEmbedding(_OptimModule):
    """z = (t, Re e^{-i f w}, Im e^{-i f w}) over the FULL seq_len, sliced to L on use (hyena.py:109-131)."""

    def __init__(self, emb_dim, seq_len, lr_pos_emb=1e-5, **kwargs):
        super().__init__()
        self.seq_len = seq_len
        t = torch.linspace(0, 1, seq_len)[None, :, None]
        bands = (emb_dim - 1) // 2
        pos = torch.linspace(0, seq_len - 1, seq_len)[None, :, None]
        w = 2 * math.pi * pos / seq_len
        f = torch.linspace(1e-4, bands - 1, bands)[None, None]
        z = torch.exp(-1j * f * w)
        self.register("z", torch.cat([t, z.real, z.imag], dim=-1), lr=lr_pos_emb)
        self.register("t", t, lr=0.0)

    def forward(self, L):
        return self.z[:, :L], self.t[:, :L]


class ExponentialModulation(_OptimModule):
    """x * (exp(-t |deltas|) + shift), deltas log-spaced decay rates per channel (hyena.py:134-155)."""

    def __init__(self, d_model, fast_decay_pct=0.3, slow_decay_pct=1.5, target=1e-2, modulation_lr=0.0,
                 shift: float = 0.0, **kwargs):
        super().__init__()
        self.shift = shift
        hi = math.log(target) / fast_decay_pct
        lo = math.log(target) / slow_decay_pct
        self.register("deltas", torch.linspace(lo, hi, d_model)[None, None], lr=modulation_lr)

    def forward(self, t, x):
        return x * (torch.exp(-t * self.deltas.abs()) + self.shift)


class HyenaFilter(_OptimModule):
    """Implicit long filter: positional embedding -> sine MLP -> exponential modulation (hyena.py:158-267).

    ``forward(x, L, k=None, bias=None)`` applies the long convolution through the HIP op regardless of
    ``fused_fft_conv`` (kept only as an accepted keyword): this package has no unfused path.
    """

    def __init__(self, d_model, emb_dim=3, order=16, fused_fft_conv=False, seq_len=1024, lr=1e-3, lr_pos_emb=1e-5,
                 dropout=0.0, w=1, wd=0, bias=True, num_inner_mlps=2, linear_mixer=False, modulate: bool = True,
                 normalized=False, bidirectional=False, **kwargs):
        super().__init__()
        self.d_model = d_model
        self.emb_dim = emb_dim
        self.seq_len = seq_len
        self.modulate = modulate
        self.use_bias = bias
        self.fused_fft_conv = fused_fft_conv
        self.bias = nn.Parameter(torch.randn(self.d_model))
        self.dropout = nn.Dropout(dropout)
        self.bidirectional = bidirectional
        self.normalized = normalized

        assert emb_dim % 2 != 0 and emb_dim >= 3, \
            "emb_dim must be odd and greater or equal to 3 (time, sine and cosine)"
        act = Sin(dim=order, w=w)          # ONE instance shared by every activation slot (hyena.py:199-213)
        self.pos_emb = PositionalEmbedding(emb_dim, seq_len, lr_pos_emb)
        if linear_mixer is False:
            layers = [nn.Linear(emb_dim, order), act]
            for _ in range(num_inner_mlps):
                layers += [nn.Linear(order, order), act]
            layers.append(nn.Linear(order, d_model, bias=False))
        else:
            layers = [nn.Linear(emb_dim, d_model, bias=False)]
        self.implicit_filter = nn.Sequential(*layers)
        self.modulation = ExponentialModulation(d_model, **kwargs)
        for child in self.implicit_filter.children():
            for name, _ in child.state_dict().items():
                getattr(child, name)._optim = {"weight_decay": wd, "lr": lr}

    def filter(self, L, *args, **kwargs):
        z, t = self.pos_emb(L)
        h = self.implicit_filter(z)
        if self.modulate:
            h = self.modulation(t, h)
        if self.normalized:
            h = h / torch.norm(h, dim=-1, p=1, keepdim=True)
        return h

    def filter_dl(self, L):
        """The same filter as ``filter(L)[0].T`` but produced directly in the (D, L) layout the long convolution
        reads (channel rows contiguous along L): the last linear layer is applied as ``W @ hidden^T`` and the
        modulation runs on (D, L), so no (L, D) -> (D, L) transposing copy (8 ms per layer at L = 2^20) is made."""
        z, t = self.pos_emb(L)
        # index the Sequential: .children() de-duplicates the ONE Sin instance that sits in three slots
        layers = [self.implicit_filter[i] for i in range(len(self.implicit_filter))]
        if self._fused_filter_ok(L, layers, z):
            lin = layers[0::2]
            mod = self.modulation
            return hyena_filter_dl(z[0], t.reshape(-1), lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias,
                                   lin[2].weight, lin[2].bias, lin[3].weight, layers[1].freq.reshape(-1),
                                   mod.deltas.reshape(-1), mod.shift, self.modulate)
        last = layers[-1]
        h = z
        for layer in layers[:-1]:
            h = layer(h)
        if not isinstance(last, nn.Linear) or last.bias is not None or h.dim() != 3 or h.shape[0] != 1:
            return self.filter(L)[0].transpose(0, 1)
        k = torch.matmul(last.weight.to(h.dtype), h[0].transpose(0, 1))          # (D, order) @ (order, L) -> (D, L)
        if self.modulate:
            deltas = self.modulation.deltas.reshape(-1, 1).abs()                   # (D, 1)
            k = k * (torch.exp(-t.reshape(1, -1) * deltas) + self.modulation.shift)
        if self.normalized:
            k = k / torch.norm(k, dim=0, p=1, keepdim=True)
        return k

    def filter_dl_split(self, L, n):
        """``filter_dl(L)`` for a filter of D n channels in the reference's '(v o)' order (hyena.py:408: channel v n + o belongs to convolution o
        of an operator of order n + 1) as n tensors (D, L), one per convolution.  With the fused filter kernels: one launch chain per convolution
        over the rows o, o + n, ... of the last linear layer and of the decay rates -- each result is written in the layout its convolution reads,
        nothing is gathered or split afterwards, and the kernels' 64 / 128 / 256 output channels are d_model instead of d_model (order - 1); the
        inner layers are evaluated n times (64 x 64 products: < 10 % of a call).  Otherwise: row views of filter_dl(L)."""
        if n == 1:
            return [self.filter_dl(L)]
        z, t = self.pos_emb(L)
        layers = [self.implicit_filter[i] for i in range(len(self.implicit_filter))]
        lin = layers[0::2]
        # short filters are launch-bound (two launch chains cost 0.4 ms more than one at 1023 x 128, profiles/r6t_order3.txt): there, one call over
        # all D n channels where the kernels take that width, and row views of it
        one_call = L <= 8192 and self._fused_filter_ok(L, layers, z)
        if not one_call and self._fused_filter_ok(L, layers, z, out_channels=lin[-1].out_features // n):
            mod = self.modulation
            deltas = mod.deltas.reshape(-1)
            return [hyena_filter_dl(z[0], t.reshape(-1), lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias, lin[2].weight, lin[2].bias,
                                    lin[3].weight[o::n], layers[1].freq.reshape(-1), deltas[o::n], mod.shift, self.modulate) for o in range(n)]
        k = self.filter_dl(L)
        k = k.reshape(k.shape[0] // n, n, k.shape[-1])
        return [k[:, o] for o in range(n)]

    def _fused_filter_ok(self, L, layers, z, out_channels=None):
        """The fused HIP filter kernels (include/hyena_filter.h) cover exactly the HyenaDNA filter configuration."""
        if len(layers) != 7:
            return False
        if not z.is_cuda and _fused_filter_requires_gpu():       # host tensors: only the test double has a "device" there
            return False
        lin, act = layers[0::2], layers[1::2]
        if not all(isinstance(m, nn.Linear) for m in lin) or not all(m is act[0] for m in act) or not isinstance(act[0], Sin):
            return False
        if any(m.bias is None for m in lin[:3]) or lin[3].bias is not None:
            return False
        if isinstance(self.modulation.deltas, nn.Parameter) and self.modulation.deltas.requires_grad and torch.is_grad_enabled():
            return False
        return fused_filter_ok(L, z.shape[-1], lin[0].out_features, out_channels or lin[3].out_features, 2, self.normalized, False)

    def forward(self, x, L, k=None, bias=None, *args, **kwargs):
        if k is None:
            k = self.filter(L)
        k = k[0] if type(k) is tuple else k
        if bias is None:
            bias = self.bias
        bias = bias if self.use_bias else 0 * bias
        if self.bidirectional:
            # hyena.py:67-73 (README "Experimental"): the input centred in the 2L window = the causal result delayed by L // 2
            y = fftconv_ref(x, k, bias.to(dtype=torch.float32), dropout_mask=None, gelu=False, bidirectional=True)
        else:
            y = fftconv_func(x, k, bias.to(dtype=torch.float32), dropout_mask=None, gelu=False,
                             force_fp16_output=torch.is_autocast_enabled())
        return y.to(dtype=x.dtype)


def _lib_max_l():
    """the longest sequence the fused core's kernels take; beyond it the generic path below runs, whose long convolution
    (fftconv_func) splits into half-length calls"""
    from . import _lib
    return _lib.MAX_L


class HyenaOperator(nn.Module):
    """Hyena operator (hyena.py:270-448): in_proj -> short depthwise conv -> order-N gated long-conv
    recurrence -> out_proj.  ``forward(u)``: (B, L, D) -> (B, min(L, l_max), D)."""

    def __init__(self, d_model, l_max, order=2, filter_order=64, num_heads=1, inner_factor=1, num_blocks=1,
                 fused_bias_fc=False, outer_mixing=False, dropout=0.0, filter_dropout=0.0, filter_cls="hyena-filter",
                 post_order_ffn=False, jit_filter=False, short_filter_order=3, activation="id", return_state=False,
                 **filter_args):
        super().__init__()
        assert d_model % num_heads == 0, f"Model dimension {d_model} must be divisible by num heads {num_heads}"
        assert l_max % num_blocks == 0, \
            f"Maximum signal length {l_max} must be divisible by block dimension {num_blocks}"
        assert order >= 2, f"Order must be at least 2, (got {order})"
        if fused_bias_fc:
            raise ImportError("fused_dense is not installed")          # as hyena.py:347-348 without flash_attn
        if filter_cls not in ("hyena-filter", HyenaFilter):
            raise NotImplementedError(f"filter_cls={filter_cls!r}: only 'hyena-filter' is provided")
        if activation not in ("id", "identity", "linear", None):
            raise NotImplementedError(f"activation={activation!r}: HyenaDNA uses 'id' (hyena.py:288)")
        if jit_filter:
            raise NotImplementedError("jit_filter references an undefined attribute in the reference (hyena.py:382)")
        self.d_model = d_model
        self.order = order
        self.l_max = l_max
        self.num_heads = num_heads
        self.inner_factor = inner_factor
        self.block_dim = l_max // num_blocks
        self.head_dim = d_model // num_heads
        self.filter_order = filter_order
        self.post_order_ffn = post_order_ffn
        self.short_filter_order = short_filter_order
        self.num_blocks = num_blocks
        self.filter_dropout = filter_dropout
        self.jit_filter = jit_filter
        self.outer_mixing = outer_mixing
        self.activation = nn.Identity()
        self.return_state = return_state
        self.dropout = nn.Dropout(dropout)

        # projections (hyena.py:345-357)
        self.out_proj = nn.Linear(d_model * inner_factor, d_model)
        self.in_proj = nn.Linear(d_model, (order + 1) * d_model)
        if post_order_ffn:
            self.ord_proj_w = nn.Parameter(torch.randn(order, num_heads, num_heads) / math.sqrt(self.head_dim))
        # filters (hyena.py:359-382)
        width = d_model * inner_factor * (order + 1)
        self.short_filter = nn.Conv1d(width, width, short_filter_order, groups=width, padding=short_filter_order - 1)
        self.filter_fn = HyenaFilter(self.head_dim * inner_factor * (order - 1), order=filter_order, seq_len=l_max,
                                     channels=1, dropout=filter_dropout, **filter_args)

    def _shell_ok(self):
        """what the fused shell kernels cover besides the order: one head, one block, inner factor 1, no outer mixing / post-order FFN, 3 short taps"""
        return (self.num_heads == 1 and self.num_blocks == 1 and self.inner_factor == 1
                and not self.outer_mixing and not self.post_order_ffn and self.short_filter_order == 3
                and (self.dropout.p == 0.0 or not self.training) and not getattr(self.filter_fn, "bidirectional", False))

    def _fused_ok(self):
        """The fused HIP mixer core covers exactly the HyenaDNA operator configuration."""
        return self.order == 2 and self._shell_ok()

    def _route(self, l):
        """which implementation ``forward`` takes for a length-l input: "fused" (order 2: the HyenaDNA configuration), "order_n" (order >= 3 on
        the same kernels, channel-major) or "generic" (the reference's graph around the HIP convolution)"""
        l_filter = min(l, self.l_max)
        if self._fused_ok() and l_filter <= _lib_max_l():
            return "fused"
        if (self.order >= 3 and ORDER_N_FUSED and CHANNEL_MAJOR and self._shell_ok() and l_filter <= _lib_max_l()
                and isinstance(self.activation, nn.Identity)):
            return "order_n"
        return "generic"

    def forward_add_norm(self, u, residual, norm_weight, norm_bias, eps):
        """``forward(u)`` followed by the prenorm block's ``residual' = out + residual; hidden = LayerNorm(residual')`` (flash_attn Block with
        fused_dropout_add_ln and dropout 0: simple_lm.py:280-284, long_conv_lm.py:381-396) in ONE pass: out_proj's matrix-core kernel forms both
        from its accumulators and the out_proj output is never written (csrc/proj_kernels.h, round 5).  Returns (hidden, residual' fp32), or None
        when the call is not served this way (the caller then runs ``forward`` and its own add + LayerNorm -- same values, bit for bit)."""
        l = u.size(-2)
        l_filter = min(l, self.l_max)
        if not (CHANNEL_MAJOR and ADD_NORM_FUSED is not False and self._fused_ok() and 0 < l_filter <= _lib_max_l() and l_filter == l and u.shape[0] > 0
                and not self.return_state and norm_weight is not None and norm_bias is not None
                and (residual is None or residual.shape == u.shape)):
            return None
        dt = torch.get_autocast_dtype("cuda" if u.is_cuda else "cpu") if torch.is_autocast_enabled() else u.dtype
        if dt not in (torch.bfloat16, torch.float16):
            return None
        if not (u.is_cuda or not _fused_filter_requires_gpu()) or not _add_norm_fused(self.d_model):
            return None
        fside = _FilterOnSideStream(u, l_filter)
        k = fside.run(lambda: self.filter_fn.filter_dl(l_filter))
        fb = self.filter_fn.bias if self.filter_fn.use_bias else 0 * self.filter_fn.bias
        xT, vg = in_proj_pre_cm(u, self.in_proj.weight, self.in_proj.bias, self.short_filter.weight, self.short_filter.bias, l_filter)
        fside.join(k)
        if not mixer_out_supported(xT, l_filter, self.out_proj.weight):
            # (xT is already made: finish on the unfused route so that nothing is computed twice)
            zT = hyena_mixer_core_cm(xT, self.in_proj.bias, self.short_filter.weight, self.short_filter.bias, k, fb, l_filter, vg=vg)
            y = out_proj_cm(zT, self.out_proj.weight, self.out_proj.bias)
            from .block import dropout_add_layer_norm
            return dropout_add_layer_norm(y, residual, norm_weight, norm_bias, 0.0, eps, prenorm=True, residual_in_fp32=True)
        return hyena_mixer_out_cm(xT, self.in_proj.bias, self.short_filter.weight, self.short_filter.bias, k, fb, l_filter, vg,
                                  self.out_proj.weight, self.out_proj.bias, add_norm=(residual, norm_weight, norm_bias, eps))

    def forward(self, u, *args, **kwargs):
        l = u.size(-2)
        l_filter = min(l, self.l_max)
        if self._fused_ok() and l_filter <= _lib_max_l():
            fside = _FilterOnSideStream(u, l_filter)
            k = fside.run(lambda: self.filter_fn.filter_dl(l_filter))           # (D, l), rows contiguous along l; on the second stream, next to in_proj
            fb = self.filter_fn.bias if self.filter_fn.use_bias else 0 * self.filter_fn.bias
            if CHANNEL_MAJOR:
                # x^T = W_in u^T straight out of the GEMM (3D, B, L): nothing between the projections is ever transposed
                # (16-bit operands at d_model 128 / 256: this package's MFMA kernel, which also hands back the conv's input v * x1
                # from its epilogue -- csrc/proj_kernels.h; otherwise the library GEMM and vg = None)
                xT, vg = in_proj_pre_cm(u, self.in_proj.weight, self.in_proj.bias, self.short_filter.weight, self.short_filter.bias,
                                        l_filter)
                fside.join(k)
                if l_filter > 0 and xT.shape[1] > 0 and mixer_out_supported(xT, l_filter, self.out_proj.weight):
                    # out_proj as this package's matrix-core kernel, the `* x0` gate on its operand load (csrc/proj_kernels.h, round 4)
                    y = hyena_mixer_out_cm(xT, self.in_proj.bias, self.short_filter.weight, self.short_filter.bias, k, fb, l_filter,
                                           vg, self.out_proj.weight, self.out_proj.bias)
                else:
                    zT = hyena_mixer_core_cm(xT, self.in_proj.bias, self.short_filter.weight, self.short_filter.bias, k, fb, l_filter,
                                             vg=vg)
                    y = out_proj_cm(zT, self.out_proj.weight, self.out_proj.bias)   # activation is the identity (_fused_ok)
            else:
                x = hyena_linear(u, self.in_proj.weight, self.in_proj.bias)     # (B, L, 3D), hipBLASLt GEMM
                z = hyena_mixer_core(x, self.short_filter.weight, self.short_filter.bias, k, fb, l_filter)
                y = hyena_linear(self.activation(z), self.out_proj.weight, self.out_proj.bias)
            return (y, None) if self.return_state else y
        if self._route(l) == "order_n":
            fside = _FilterOnSideStream(u, l_filter)
            ks = fside.run(lambda: self.filter_fn.filter_dl_split(l_filter, self.order - 1))   # one (D, l) filter per convolution ('(v o)' channels, hyena.py:408)
            fb = self.filter_fn.bias if self.filter_fn.use_bias else 0 * self.filter_fn.bias
            xT = in_proj_cm(u, self.in_proj.weight)                             # ((order + 1) D, B, L), bias added on load by the shell kernels
            fside.join(ks)
            zT = hyena_mixer_core_cm_order_n(xT, self.in_proj.bias, self.short_filter.weight, self.short_filter.bias, ks, fb, l_filter, self.order)
            y = out_proj_cm(zT, self.out_proj.weight, self.out_proj.bias)
            return (y, None) if self.return_state else y
        u = self.in_proj(u).transpose(1, 2)                                     # b l d -> b d l
        uc = self.short_filter(u)[..., :l_filter]
        b = uc.shape[0]
        h, z, o1 = self.num_heads, self.num_blocks, self.order + 1
        uc = uc.reshape(b, h, self.head_dim * o1, z, l_filter // z)             # b (ho v) (z l) -> b ho v z l
        *x, v = uc.split(self.d_model, dim=2)
        k = self.filter_fn.filter(l_filter)                                     # (1, l, (v o))
        k = k.reshape(k.shape[0], l_filter, self.head_dim, self.order - 1).permute(0, 3, 2, 1)[0]   # o v l
        bias = self.filter_fn.bias.reshape(self.head_dim, self.order - 1).transpose(0, 1)            # o v
        for o, x_i in enumerate(reversed(x[1:])):
            if self.outer_mixing:
                v = self.dropout(v.unsqueeze(2) * x_i.unsqueeze(3)).sum(dim=2)
            else:
                v = self.dropout(v * x_i)
            v = self.filter_fn(v, l_filter, k=k[o], bias=bias[o, None, :, None])
            if self.post_order_ffn:
                w = self.ord_proj_w[o]
                v = (w[None, :, :, None, None, None] * v.unsqueeze(2)).sum(dim=1)
        y = (v * x[0]).permute(0, 3, 4, 1, 2).reshape(b, l_filter, h * self.head_dim)   # b h v z l -> b (z l) (h v)
        y = self.out_proj(self.activation(y))
        if self.return_state:
            return y, None
        return y

    @property
    def d_output(self):
        return self.d_model
