"""Python op seam of the hot path: the MI355X-native counterpart of the reference's ``src/ops/fftconv.py``.

Exports the names ``src/models/sequence/hyena.py:12-16`` imports -- ``fftconv_func``, ``fftconv_ref``,
``fftconv_heads_ref`` -- plus the autograd function ``FFTConvFunc`` (reference: ``src/ops/fftconv.py:58-108``),
all backed by the HIP kernels behind the C ABI of ``include/hyena_fftconv.h``.

Semantics (reference ``fftconv_ref``, ``src/models/sequence/hyena.py:59-88``, the function every HyenaDNA config
runs): ``out = irfft(rfft(u, 2L) * rfft(k, 2L) / 2L, norm='forward')[..., :L] + u * D[..., None]`` with all FFT
math in fp32 and the result cast back to ``u.dtype``.

Differences from the reference's fused op (all lifts of restrictions):
* ``u`` may be 3-D ``(B, H, L)`` or the 5-D ``(b, h, v, z, l)`` tensor ``HyenaOperator`` really passes
  (``hyena.py:396-423``) with ``k`` ``(v, l)`` and ``D`` ``(1, v, 1)`` / ``(v,)``; any ``L`` in ``[1, 2**20]``.
* saves only ``(u, k, D)`` for backward and recomputes spectra (the reference's autograd keeps ``u_f``,
  8(L+1) bytes per row).
* ``k_rev`` (src/ops/fftconv.py:64-66: ``k_f + conj(rfft(k_rev))`` = an anti-causal convolution added to the causal one) and
  ``bidirectional`` (hyena.py:67-73: the input centred in the 2L window = the causal result delayed by L // 2) are served by the same
  kernels through flips / shifts (``fftconv_func`` / ``fftconv_ref`` below);
* the H3-form options of the reference op that no HyenaDNA config enables -- ``gelu``, ``dropout_mask``, ``v`` / ``q`` / ``head_dim``,
  ``output_hbl_layout``, ``force_fp16_output``, ``fftfp16`` (src/ops/fftconv.py:37-55, 58-108; the kernel's order of operations
  csrc/fftconv/fftconv_cuda.cu:420-500) -- are served by ``fftconv_func`` since round 4: the long convolution runs on the HIP kernels
  with fp32 rows, the element-wise parts around it (the ``k (x) v`` outer product, GELU, the dropout mask, ``* q`` and the sum over the
  head dimension) are PyTorch ops on the fp32 result, rounded once at the end like the fused reference kernel (``fftfp16`` asks the
  reference for a faster, less exact half-precision FFT: here the transform stays fp32).  ``FFTConvFunc.apply`` itself takes the plain
  form only and says so;
* sequences longer than the kernels' 2^20 positions (the reference's torch.fft path takes any L) are served by splitting the causal
  convolution into four half-length ones (``_conv_long``); slower than a native transform of that size would be, but exact.
There is no CPU / torch.fft fallback in this module.
"""
import torch

from . import _gradmode, _lib

__all__ = ["fftconv_func", "FFTConvFunc", "fftconv_ref", "fftconv_heads_ref"]


def _unsupported(**opts):
    bad = [name for name, on in opts.items() if on]
    if bad:
        raise NotImplementedError(
            "hyena_dna_amd.fftconv: option(s) %s are not implemented by the MI355X kernel "
            "(no HyenaDNA configuration enables them; reference: src/ops/fftconv.py:60-61)" % ", ".join(bad))


def _as_rows(u, n_channels):
    """View/copy ``u`` as a contiguous (B', D, L) tensor whose dim 1 is the filter's channel axis.

    3-D (B, D, L): as is.  5-D (b, h, v, z, l) as passed by HyenaOperator: channel axis is ``v``
    (k_f.unsqueeze(1) in hyena.py:77-78 broadcasts the filter over b, h and z).
    Returns (rows, restore) with restore(t) mapping a (B', D, L) tensor back to u's shape.
    """
    if u.dim() == 3:
        if u.shape[1] != n_channels:
            raise ValueError(f"u has {u.shape[1]} channels but k has {n_channels}")
        return u.contiguous(), (lambda t: t)
    if u.dim() == 5:
        b, h, v, z, l = u.shape
        if v != n_channels:
            raise ValueError(f"u has {v} channels (dim 2) but k has {n_channels}")
        if z == 1:
            return u.contiguous().view(b * h, v, l), (lambda t: t.view(b, h, v, z, l))
        rows = u.permute(0, 1, 3, 2, 4).contiguous().view(b * h * z, v, l)
        return rows, (lambda t: t.view(b, h, z, v, l).permute(0, 1, 3, 2, 4))
    raise ValueError(f"fftconv expects a 3-D (B, H, L) or 5-D (b, h, v, z, l) input, got shape {tuple(u.shape)}")


def _aligned(t):
    return t if t.data_ptr() % 16 == 0 else t.clone()


class FFTConvFunc(torch.autograd.Function):
    """Autograd contract of the fused op (reference: src/ops/fftconv.py:58-103): grads for (u, k, D)."""

    @staticmethod
    def forward(ctx, u, k, D, dropout_mask=None, gelu=True, force_fp16_output=False, output_hbl_layout=False,
                v=None, head_dim=1, q=None, fftfp16=False, k_rev=None):
        _unsupported(dropout_mask=dropout_mask is not None, gelu=bool(gelu), output_hbl_layout=bool(output_hbl_layout),
                     v=v is not None, q=q is not None, head_dim=head_dim != 1, fftfp16=bool(fftfp16),
                     k_rev=k_rev is not None)          # (k_rev: composed from two calls by fftconv_func)
        if k.dim() != 2:
            raise ValueError(f"k must be (H, L), got {tuple(k.shape)}")
        H, L = k.shape
        if u.shape[-1] != L:
            raise ValueError(f"u has sequence length {u.shape[-1]} but k has {L}")
        rows, restore = _as_rows(u, H)
        rows = _aligned(rows)
        kf = _aligned(k.detach().to(torch.float32).contiguous())
        bias = None
        if D is not None:
            if D.numel() != H:
                raise ValueError(f"D must have {H} elements, got shape {tuple(D.shape)}")
            bias = D.detach().to(torch.float32).reshape(H).contiguous()
        # keep the forward's column spectra for the backward when any gradient is wanted (time-for-memory trade,
        # _lib.save_spectra_default); otherwise the backward recomputes them from (u, k)
        want_grad = any(_gradmode.needs(ctx)[:3])
        saved = None
        if want_grad and _lib.save_spectra_default(*rows.shape, device=rows.device):
            out, saved = _lib.fftconv_fwd(rows, kf, bias, save=True)
        else:
            out = _lib.fftconv_fwd(rows, kf, bias, grad=want_grad)
        ctx.spectra = saved
        ctx.save_for_backward(rows, kf, bias if bias is not None else torch.empty(0, device=u.device))
        ctx.has_bias = bias is not None
        ctx.restore = restore
        ctx.k_dtype = k.dtype
        ctx.D_meta = (D.shape, D.dtype) if D is not None else None
        out = restore(out)
        if force_fp16_output and u.dtype == torch.float32:      # fftconv.cpp:110 (bf16 input overrides it)
            out = out.to(torch.float16)
        return out

    @staticmethod
    def backward(ctx, dout):
        rows, kf, bias = ctx.saved_tensors
        bias = bias if ctx.has_bias else None
        H = kf.shape[0]
        g, _ = _as_rows(dout.to(rows.dtype), H)
        g = _aligned(g)
        need_du = ctx.needs_input_grad[0]
        need_dk = ctx.needs_input_grad[1] or (ctx.needs_input_grad[2] and ctx.has_bias)
        du, dk, dbias = _lib.fftconv_bwd(g, rows, kf, bias, need_du=need_du, need_dk=need_dk, saved=ctx.spectra)
        ctx.spectra = None
        du = ctx.restore(du) if (du is not None and need_du) else None
        dk_out = dk.to(ctx.k_dtype) if (dk is not None and ctx.needs_input_grad[1]) else None
        dD = None
        if ctx.needs_input_grad[2] and ctx.D_meta is not None:
            shape, dtype = ctx.D_meta
            dD = dbias.reshape(shape).to(dtype)
        return du, dk_out, dD, None, None, None, None, None, None, None, None, None


def _conv(u, k, D, dropout_mask, gelu, force_fp16_output, output_hbl_layout, v, head_dim, q, fftfp16):
    if u.shape[-1] > _lib.MAX_L and not (dropout_mask is not None or gelu or output_hbl_layout or v is not None or q is not None
                                         or head_dim != 1 or fftfp16):
        out = _conv_long(u, k, D)
        return out.to(torch.float16) if (force_fp16_output and u.dtype == torch.float32) else out
    return _gradmode.apply(FFTConvFunc, u, k, D, dropout_mask, gelu, force_fp16_output, output_hbl_layout, v, head_dim, q, fftfp16, None)


def _plain(u, k, D):
    """causal convolution + D u on the HIP kernels, any length, plain form"""
    return _conv(u, k, D, None, False, False, False, None, 1, None, False)


def _conv_long(u, k, D):
    """L > 2^20 (the kernels' largest transform; the reference's torch.fft path, hyena.py:61, takes any L): with u = [u_lo | u_hi],
    k = [k_lo | k_hi] split at h = ceil(L / 2), and C(a, b) the causal convolution of two length-h rows truncated to h outputs (the
    kernels' primitive, any h <= 2^20; longer halves recurse),

        out[:h] = C(u_lo, k_lo) + D u_lo
        out[h:] = upper(u_lo * k_lo) + C(u_lo, k_hi) + C(u_hi, k_lo) + D u_hi

    where upper(a * b)[n] = (a * b)[h + n] -- the half of the linear convolution that C drops -- is itself a causal convolution of the
    flipped rows read backwards: upper[n] = C(flip a, flip b)[h - 2 - n] (and 0 at n = h - 1).  Four half-length calls, i.e. twice the
    work of one transform of the full length, all through FFTConvFunc (gradients come from autograd over this composition); the halves are
    convolved as fp32 rows and the sum is rounded once, like the reference's single irfft."""
    L = u.shape[-1]
    h = (L + 1) // 2
    uf = u.float()
    kf = k.float()
    if L % 2:                                              # odd L: one zero behind each row; the extra output is dropped
        uf = torch.nn.functional.pad(uf, (0, 1))
        kf = torch.nn.functional.pad(kf, (0, 1))
    u_lo, u_hi = uf[..., :h].contiguous(), uf[..., h:].contiguous()
    k_lo, k_hi = kf[..., :h].contiguous(), kf[..., h:].contiguous()
    lo = _plain(u_lo, k_lo, D)
    w = _plain(u_lo.flip(-1), k_lo.flip(-1), None)                             # C(flip u_lo, flip k_lo)
    upper = torch.nn.functional.pad(w.flip(-1)[..., 1:], (0, 1))               # upper[n] = w[h - 2 - n], upper[h - 1] = 0
    hi = upper + _plain(u_lo, k_hi, None) + _plain(u_hi, k_lo, D)
    return torch.cat([lo, hi], dim=-1)[..., :L].to(u.dtype)


def fftconv_func(u, k, D, dropout_mask=None, gelu=True, force_fp16_output=False, output_hbl_layout=False, v=None,
                 head_dim=1, q=None, fftfp16=False, k_rev=None):
    """Same signature as the reference's ``fftconv_func`` (src/ops/fftconv.py:105-108).

    ``k_rev`` (src/ops/fftconv.py:64-66; hyena.py:63-65): the reference adds ``conj(rfft(k_rev, 2L))`` to the filter spectrum, i.e. the
    circularly time-reversed ``k_rev``: ``out[t] += sum_{s >= t} k_rev[s - t] u[s]`` -- an anti-causal convolution, which is the causal
    one of the flipped input, flipped back.  Both run on the HIP kernels; 16-bit inputs are convolved in fp32 so that the sum is
    rounded once, as in the reference.

    The H3 form (``v`` and ``q`` given; src/ops/fftconv.py:37-55, fftconv_cuda.cu:405-500): with ``u`` (b, h d1, l), ``v`` (b, h d2, l),
    ``q`` (b, h d1, l), ``k`` (h, l), ``D`` (h,):  kv = u (x) v over (d1, d2);  y = conv(kv, k) + D kv;  [GELU];  out = sum_d1 y q  ->
    (b, h d2, l).  ``dropout_mask`` (b, H) scales the rows after the GELU (plain form).  ``output_hbl_layout`` returns the same (b, h, l)
    tensor laid out as (h, b, l) in memory, as the reference kernel writes it."""
    plain = (dropout_mask is None and not gelu and not output_hbl_layout and v is None and q is None and head_dim == 1 and not fftfp16)
    if plain and k_rev is None:
        return _conv(u, k, D, None, False, force_fp16_output, False, None, 1, None, False)
    if v is None and q is not None:
        raise ValueError("fftconv_func: q without v (the reference kernel selects the H3 form on v alone, fftconv_cuda.cu:806, and would "
                         "silently ignore q)")
    # v WITHOUT q: the reference op accepts the call (src/ops/fftconv.py:58-84 checks nothing; its kernel then reads q through
    # a null pointer).  Served with the only meaning the formula leaves: no q-multiply -- q = 1 -- so the head sum is a plain sum over d1.
    if v is None and head_dim != 1:
        raise ValueError("fftconv_func: head_dim > 1 without v / q")
    if v is not None and dropout_mask is not None:
        raise NotImplementedError("fftconv_func: dropout_mask together with the H3 form (v, q) -- no caller in the reference builds it")

    def conv32(x32):            # causal (+ anti-causal) convolution and the D term on fp32 rows
        y = _plain(x32, k, D)
        if k_rev is not None:
            y = y + _plain(x32.flip(-1), k_rev, None).flip(-1)
        return y

    if v is None:
        out = conv32(u.float())
    else:
        from einops import rearrange
        kv = (rearrange(u, "b (h d1) l -> b d1 1 h l", d1=head_dim).float()
              * rearrange(v, "b (h d2) l -> b 1 d2 h l", d2=head_dim).float())             # b d1 d2 h l   (fftconv.py:40-41)
        b, d1, d2, h, l = kv.shape
        out = conv32(kv.reshape(b * d1 * d2, h, l)).reshape(b, d1, d2, h, l)               # y + kv D      (fftconv.py:48-49)
    if gelu:
        out = torch.nn.functional.gelu(out)                                                # fftconv_cuda.cu:473
    if dropout_mask is not None:
        out = out * dropout_mask.to(out.dtype).unsqueeze(-1)                               # (b, H) -> rows
    if v is not None:
        from einops import rearrange
        if q is not None:
            out = out * rearrange(q, "b (h d1) l -> b d1 1 h l", d1=head_dim).float()
        out = rearrange(out.sum(dim=1), "b d2 h l -> b (h d2) l")                          # fftconv.py:50-55
    odt = torch.float16 if (force_fp16_output and u.dtype == torch.float32) else u.dtype
    out = out.to(odt)
    if output_hbl_layout:                                                                  # (b, h, l) values, (h, b, l) memory order
        out = out.transpose(0, 1).contiguous().transpose(0, 1)
    return out


def fftconv_ref(u, k, D, dropout_mask=None, gelu=True, k_rev=None, bidirectional=False):
    """Name-compatible with the reference's ``fftconv_ref`` (src/ops/fftconv.py:15-34; hyena.py:59-88), but
    routed through the same HIP kernels: this package ships no torch.fft implementation.

    ``bidirectional`` (hyena.py:67-73): the reference centres the input in the 2L-point window (pb = L // 2 zeros in front) before the
    CIRCULAR product with the L-tap filter and keeps the first L outputs.  Written out, with c = L - pb + 1:
        out[t] = sum_{j <= t - pb} k[t - pb - j] u[j]            the causal convolution, delayed by pb positions
               + sum_{m >= 0} k[L - 1 - m] u[t + c + m]          the wrap-around: later inputs through the filter's tail
               + D u[t]
    i.e. one causal convolution, and one anti-causal convolution (= the causal one of the flipped input, flipped back) of the
    input advanced by c positions with the flipped filter -- two calls of the same HIP kernels.  Together with ``k_rev`` (a
    combination hyena.py never builds: hyena.py:261 passes ``bidirectional`` alone) it is not implemented."""
    if not bidirectional:
        return fftconv_func(u, k, D, dropout_mask=dropout_mask, gelu=gelu, k_rev=k_rev)
    _unsupported(dropout_mask=dropout_mask is not None, gelu=bool(gelu), k_rev_with_bidirectional=k_rev is not None)
    pad = torch.nn.functional.pad
    L = u.shape[-1]
    pb = L // 2
    c = L - pb + 1
    uf = u.float()
    y = pad(fftconv_func(uf, k, None, gelu=False)[..., : L - pb], (pb, 0))
    if c < L:
        adv = pad(uf[..., c:], (0, c))                                      # u advanced by c positions, zeros behind
        y = y + fftconv_func(adv.flip(-1), k.flip(-1), None, gelu=False).flip(-1)
    if D is not None:
        Dv = D if D.dim() > 1 else D.unsqueeze(-1)                     # (1, H, 1) as HyenaOperator passes it, or (H,)
        if u.dim() == 5 and D.dim() == 3:
            Dv = D.reshape(1, 1, -1, 1, 1)
        y = y + uf * Dv
    return y.to(u.dtype)


def fftconv_heads_ref(*args, **kwargs):
    """Imported by the reference's hyena.py:13 but defined nowhere in the reference; it only has
    to exist for that import to succeed."""
    raise NotImplementedError("fftconv_heads_ref does not exist in the reference either (hyena.py:13)")
