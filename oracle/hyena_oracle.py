"""CPU oracle for the Hyena long-convolution hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  Nothing under ``hyena_dna_amd/`` imports it; the product path has no
CPU fallback and raises when the HIP library is missing.

It restates, function by function, the arithmetic of the reference's *pure PyTorch* path
(the path every HyenaDNA config actually runs -- SURVEY.md section 0.1).  Citations are
into ``/root/reference``:

* ``fftconv_ref``           <- src/models/sequence/hyena.py:59-88
                               (dup: src/ops/fftconv.py:15-34, standalone_hyenadna.py:45-60)
* ``fftconv_h3_ref``        <- src/ops/fftconv.py:37-55 (the H3 form of the fused op: k (x) v, SSM-kernel convolution, q gate;
                               pinned to the reference's own function by tests/test_overlay_reference.py)
* ``positional_embedding``  <- src/models/sequence/hyena.py:109-131
* ``filter_mlp`` / ``Sin``  <- src/models/sequence/hyena.py:96-106, 199-215
* ``exp_modulation``        <- src/models/sequence/hyena.py:134-155
* ``hyena_filter``          <- src/models/sequence/hyena.py:229-238
* ``hyena_filter_autocast`` <- the same under ``torch.autocast`` (the trainer's ``precision: 16`` -- "bf16 only a100" --,
                               configs/experiment/hg38/hg38_hyena.yaml:41): PyTorch's own autocast
                               does the casting; pinned to the reference's HyenaFilter under CPU autocast by
                               ``oracle/make_golden_filter_autocast.py`` / ``tests/golden/hyena_filter_autocast.pt``
* ``short_conv``            <- src/models/sequence/hyena.py:363-369, 394 (nn.Conv1d, groups=C, padding=k-1, cut to L)
* ``hyena_operator``        <- src/models/sequence/hyena.py:388-444 (order-N recurrence, defaults only)
* ``causal_conv_direct_f64``   an independent O(L^2) float64 truth for small L (no FFT at all)

Parity pinning: the reference ships NO golden vectors or tests for this path (SURVEY.md
section 4, 8c), so the oracle is pinned against outputs of the reference itself, executed in
the build container by ``oracle/make_golden.py`` (which imports the real reference classes)
and committed under ``tests/golden/``.  ``tests/test_oracle_golden.py`` checks this module
against those fixtures bit-for-bit where the op sequence is identical.

The arithmetic itself lives in PyTorch (``torch.fft.rfft/irfft``, ``F.conv1d``, ``F.linear``),
a third-party dependency of the reference (pinned there as torch 1.13 + CUDA 11.7,
README.md:68-72); this image has torch 2.10 CPU kernels (pocketfft).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# the long convolution  (hyena.py:59-88)
# --------------------------------------------------------------------------------------
def fftconv_ref(u, k, D, dropout_mask=None, gelu=False, k_rev=None, bidirectional=False):
    """out = irfft(rfft(u, 2L) * rfft(k, 2L) / 2L, norm='forward')[..., :L] + u * D[..., None].

    u: (..., L) any float dtype; k: (H, L) fp32; D: broadcastable to u[..., 0].
    All FFT math runs in k.dtype (fp32), result is cast back to u.dtype (hyena.py:75, 88).
    """
    seqlen = u.shape[-1]
    fft_size = 2 * seqlen                                         # hyena.py:61
    k_f = torch.fft.rfft(k, n=fft_size) / fft_size                # hyena.py:62
    if k_rev is not None:                                         # hyena.py:63-65
        k_f = k_f + (torch.fft.rfft(k_rev, n=fft_size) / fft_size).conj()
    if bidirectional:                                             # hyena.py:67-73
        padded_length = seqlen + 2 * (seqlen // 2)
        pad_before = padded_length // 2 - (seqlen // 2)
        pad_after = padded_length - seqlen - pad_before
        u_in = F.pad(u, (pad_before, pad_after), mode="constant", value=0)
    else:
        u_in = u
    u_f = torch.fft.rfft(u_in.to(dtype=k.dtype), n=fft_size)      # hyena.py:73/75
    if u.dim() > 3:                                               # hyena.py:77-78
        k_f = k_f.unsqueeze(1)
    y = torch.fft.irfft(u_f * k_f, n=fft_size, norm="forward")[..., :seqlen]   # hyena.py:80
    out = y + u * D.unsqueeze(-1)                                 # hyena.py:82
    if gelu:
        out = F.gelu(out)                                         # hyena.py:83-84
    if dropout_mask is not None:                                  # hyena.py:85-86
        return (out * dropout_mask[..., None]).to(dtype=u.dtype)
    return out.to(dtype=u.dtype)                                  # hyena.py:88


def fftconv_h3_ref(k, ssm_kernel, D, q, v, head_dim=1, ssm_kernel_rev=None):
    """The H3 form of the fused op (src/ops/fftconv.py:37-55), restated line by line: kv = k (x) v over the head dimension, long
    convolution with the SSM kernel (+ its time reversal), the D term, then the q gate and the sum over d1.
    k, q: (b, h d1, l); v: (b, h d2, l); ssm_kernel: (h, l); D: (h,)."""
    from einops import rearrange
    seqlen = k.shape[-1]
    fft_size = 2 * seqlen                                                               # fftconv.py:39
    kv = (rearrange(k, "b (h d1) l -> b d1 1 h l", d1=head_dim)
          * rearrange(v, "b (h d2) l -> b 1 d2 h l", d2=head_dim))                       # fftconv.py:40-41
    kv_f = torch.fft.rfft(kv.to(dtype=ssm_kernel.dtype), n=fft_size) / fft_size         # fftconv.py:42
    ssm_kernel_f = torch.fft.rfft(ssm_kernel, n=fft_size)                               # fftconv.py:43
    if ssm_kernel_rev is not None:                                                      # fftconv.py:44-46
        ssm_kernel_f = ssm_kernel_f + torch.fft.rfft(ssm_kernel_rev, n=fft_size).conj()
    y = torch.fft.irfft(kv_f * ssm_kernel_f, n=fft_size, norm="forward")[..., :seqlen]  # fftconv.py:47
    out = y + kv * D.unsqueeze(-1)                                                      # fftconv.py:48
    q = rearrange(q, "b (h d1) l -> b d1 1 h l", d1=head_dim)                           # fftconv.py:49
    if head_dim > 1:                                                                    # fftconv.py:50-52
        out = (out * q).sum(dim=1)
        return rearrange(out, "b d2 h l -> b (h d2) l").to(dtype=k.dtype)
    return rearrange(out * q, "b 1 1 h l -> b h l").to(dtype=k.dtype)                   # fftconv.py:53-54


def causal_conv_direct_f64(u, k, D):
    """Independent float64 truth: y[t] = sum_{s<=t} k[s] u[t-s] + D u[t].  O(L^2); small L only.

    u: (R, L); k: (R, L); D: (R,).  No FFT, no padding subtleties.
    """
    u64, k64, D64 = u.double(), k.double(), D.double()
    R, L = u64.shape
    y = torch.zeros_like(u64)
    for s in range(L):
        y[:, s:] += k64[:, s:s + 1] * u64[:, : L - s]
    return y + u64 * D64[:, None]


# --------------------------------------------------------------------------------------
# the implicit filter  (hyena.py:96-155, 229-238)
# --------------------------------------------------------------------------------------
def positional_embedding(emb_dim: int, seq_len: int):
    """z: (1, seq_len, emb_dim), t: (1, seq_len, 1).  hyena.py:109-131."""
    t = torch.linspace(0, 1, seq_len)[None, :, None]
    bands = (emb_dim - 1) // 2
    t_rescaled = torch.linspace(0, seq_len - 1, seq_len)[None, :, None]
    w = 2 * math.pi * t_rescaled / seq_len
    f = torch.linspace(1e-4, bands - 1, bands)[None, None]
    z = torch.exp(-1j * f * w)
    z = torch.cat([t, z.real, z.imag], dim=-1)
    return z, t


def exp_modulation_deltas(d_model, fast_decay_pct=0.3, slow_decay_pct=1.5, target=1e-2):
    """hyena.py:145-149."""
    max_decay = math.log(target) / fast_decay_pct
    min_decay = math.log(target) / slow_decay_pct
    return torch.linspace(min_decay, max_decay, d_model)[None, None]


def hyena_filter(sd: Dict[str, torch.Tensor], L: int, prefix: str = "filter_fn.",
                 modulate: bool = True, shift: float = 0.0, normalized: bool = False):
    """HyenaFilter.filter(L): (1, L, d).  hyena.py:229-238 with Sin (96-106), modulation (152-155).

    ``sd`` uses the reference's state_dict names (checkpoint row).
    """
    z = sd[prefix + "pos_emb.z"][:, :L]
    t = sd[prefix + "pos_emb.t"][:, :L]
    h = z
    i = 0
    while (prefix + f"implicit_filter.{i}.weight") in sd:
        w = sd[prefix + f"implicit_filter.{i}.weight"]
        b = sd.get(prefix + f"implicit_filter.{i}.bias")
        h = F.linear(h, w, b)
        fkey = prefix + f"implicit_filter.{i + 1}.freq"
        if fkey in sd:
            h = torch.sin(sd[fkey] * h)                           # hyena.py:105-106
        i += 2
    if modulate:
        decay = torch.exp(-t * sd[prefix + "modulation.deltas"].abs())    # hyena.py:153
        h = h * (decay + shift)                                            # hyena.py:154
    if normalized:
        h = h / torch.norm(h, dim=-1, p=1, keepdim=True)          # hyena.py:235-236
    return h


# --------------------------------------------------------------------------------------
# short depthwise conv + the operator  (hyena.py:363-369, 388-444)
# --------------------------------------------------------------------------------------
def short_conv(u_bdl, weight, bias, L_out):
    """Depthwise causal conv: nn.Conv1d(C, C, k, groups=C, padding=k-1)(u)[..., :L_out]."""
    C = u_bdl.shape[1]
    ksz = weight.shape[-1]
    return F.conv1d(u_bdl, weight, bias, padding=ksz - 1, groups=C)[..., :L_out]


def short_conv_taps(u_bdl, weight, bias, L_out):
    """The same depthwise causal convolution written out tap by tap (Conv1d(C, C, k, groups=C, padding=k-1)(u)[..., :L_out]:
    y[c, t] = bias[c] + sum_j weight[c, 0, j] u[c, t - (k - 1) + j], zero left padding; hyena.py:363-369, 394).  Plain
    element-wise ops, so it runs in any dtype on any device -- F.conv1d over millions of positions in 768 groups takes the host
    minutes, and has no float64 device kernel.  tests/test_oracle_golden.py pins it to `short_conv`."""
    ksz = weight.shape[-1]
    L_in = u_bdl.shape[-1]
    n = min(L_out, L_in + ksz - 1)
    up = F.pad(u_bdl, (ksz - 1, 0))                                # (B, C, L_in + k - 1): up[..., t + j] = u[..., t - (k-1) + j]
    y = None
    for j in range(ksz):
        hi = min(n, up.shape[-1] - j)
        term = weight[None, :, 0, j, None] * up[..., j:j + hi]
        if hi < n:
            term = F.pad(term, (0, n - hi))
        y = term if y is None else y + term
    return y + bias[None, :, None]


def hyena_filter_autocast(sd: Dict[str, torch.Tensor], L: int, dtype=torch.bfloat16, **kw):
    """:func:`hyena_filter` as the reference evaluates it in training: the whole model runs under ``torch.autocast`` (Lightning's
    ``trainer.precision: 16``, configs/experiment/hg38/hg38_hyena.yaml:41), so the filter MLP's four ``nn.Linear`` run in
    the 16-bit autocast type (inputs, weights, biases, outputs rounded; fp32 accumulation) while ``Sin`` (fp32 ``freq`` times a 16-bit
    tensor) and the modulation are promoted to fp32.  PyTorch's own autocast does the casting here -- on whatever device ``sd`` lives
    on -- with the parameters first brought to fp32 (they are fp32 in the reference model); the result is cast to ``sd``'s dtype so
    that a higher-precision evaluation of the rest of the operator can consume it."""
    some = sd[kw.get("prefix", "filter_fn.") + "pos_emb.z"]
    sd32 = {k: (v.to(torch.float32) if v.is_floating_point() else v) for k, v in sd.items()}
    with torch.autocast(some.device.type, dtype=dtype):
        k = hyena_filter(sd32, L, **kw)
    return k.to(some.dtype)


def hyena_operator(sd: Dict[str, torch.Tensor], u: torch.Tensor, l_max: int, order: int = 2,
                   modulate: bool = True, shift: float = 0.0, conv_fn=None, short_conv_fn=None, filter_fn=None):
    """HyenaOperator.forward for the default options (num_heads=1, num_blocks=1, inner_factor=1,
    no outer mixing / post-order FFN, activation 'id', dropout 0).  hyena.py:388-444.

    u: (B, L, D) -> (B, L', D) with L' = min(L, l_max).  ``conv_fn`` lets a test swap the long
    conv (default: :func:`fftconv_ref`), ``filter_fn`` the filter evaluation (default: :func:`hyena_filter`;
    :func:`hyena_filter_autocast` for the training-time graph).
    """
    conv_fn = conv_fn or fftconv_ref
    filter_fn = filter_fn or hyena_filter
    short_conv_fn = short_conv_fn or short_conv
    l = u.size(-2)
    l_filter = min(l, l_max)                                       # hyena.py:389-390
    d_model = sd["out_proj.weight"].shape[0]
    x = F.linear(u, sd["in_proj.weight"], sd["in_proj.bias"])      # hyena.py:391
    x = x.transpose(1, 2)                                          # b l d -> b d l (392)
    uc = short_conv_fn(x, sd["short_filter.weight"], sd["short_filter.bias"], l_filter)   # 394
    B = uc.shape[0]
    uc = uc.reshape(B, 1, d_model * (order + 1), 1, l_filter)      # hyena.py:396-402
    *xs, v = uc.split(d_model, dim=2)                              # hyena.py:404
    k = filter_fn(sd, l_filter, modulate=modulate, shift=shift)    # (1, L, D*(order-1))
    # 'c l (v o) -> c o v l'                                        hyena.py:408
    k = k.reshape(1, l_filter, d_model, order - 1).permute(0, 3, 2, 1)[0]
    bias = sd["filter_fn.bias"].reshape(d_model, order - 1).transpose(0, 1)   # '(v o) -> o v' (410-412)
    for o, x_i in enumerate(reversed(xs[1:])):                     # hyena.py:414
        v = v * x_i                                                # hyena.py:420 (dropout p=0)
        kk = k[o]
        bb = bias[o, None, :, None]
        v = conv_fn(v, kk, bb, None, gelu=False).to(v.dtype)       # hyena.py:423, 261-267
    y = (v * xs[0]).reshape(B, d_model, l_filter).transpose(1, 2)  # 'b h v z l -> b (z l) (h v)' 432-439
    return F.linear(y, sd["out_proj.weight"], sd["out_proj.bias"]) # hyena.py:440
