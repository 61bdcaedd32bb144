"""The operator's two dense projections (``in_proj`` / ``out_proj``, ``src/models/sequence/hyena.py:350-351,391,440``)
as plain library GEMMs (hipBLASLt MFMA kernels through PyTorch) -- with one repair for HyenaDNA's shapes.

The weight gradient of ``y = x W^T + b`` with ``x`` of shape (B*L, K) is ``dy^T x``: an (N, K) product whose contraction
runs over all B*L positions.  At L = 2^20 hipBLASLt schedules it as N*K / tile = 12 workgroups on a 256-CU part
(measured 2.4 ms for K = 256, N = 768 and 1.9 ms for N = 256, against 0.45 / 0.22 ms of memory time,
``scripts/gemm_probe.py``).  ``SplitKLinearFunc`` computes the same gradient as a batched GEMM over S position slices with
fp32 partial results summed in a fixed order (0.5 / 0.25 ms; deterministic), and the bias gradient as one fp32 column
sum.  Forward and input gradient are the ordinary library GEMMs.

Used by ``hyena_dna_amd.hyena.HyenaOperator`` on the GPU when there are at least ``MIN_ROWS`` positions (16-bit autocast, or
plain fp32 / 16-bit tensors); everything else goes through ``torch.nn.functional.linear`` unchanged.
"""
import os

import torch
import torch.nn.functional as F

from . import _castcache

__all__ = ["hyena_linear", "SplitKLinearFunc", "split_count", "split_k_weight_grad", "in_proj_cm", "out_proj_cm", "in_proj_pre_cm"]

MIN_ROWS = 32768          # below this the library's own schedule is fine
MAX_SPLITS = 64
MIN_SLICE = 4096


def split_count(rows, out_elems=None):
    """Number of equal position slices of the batched weight-gradient GEMM: the largest power of two <= MAX_SPLITS with
    slices of >= MIN_SLICE rows.  ``rows`` need not be divisible by it (split_plan; the reference dataset yields L = max_length - 1 --
    999,999 / 449,999 / 159,999 / 32,767, hg38_dataset.py:220 -- so B*L is odd in real training).
    ``out_elems``: size of the gradient.  A 256 x 256 gradient (out_proj at d_model 256) is ONE output tile per slice, so 64 slices are 64
    workgroups on 256 CUs: such products take up to 256 slices (measured at 2^20 - 1 positions, scripts/wgrad_slice_probe.py: 64 x 16383 rows
    527 us, 128 x 8191 342 us, 255 x 4096 337 us; the 768 x 256 and 1024 x 256 gradients are best at 64: 552 us vs 636 at 127)."""
    cap = MAX_SPLITS * 4 if (out_elems is not None and out_elems == 256 * 256) else MAX_SPLITS       # (the measured shape only: ADVICE r5)
    s = 1
    while s < cap and rows // (2 * s) >= MIN_SLICE - 64:           # (- 64: 8 x 32767 rows are cut like 8 x 32768 ones, not in half as many slices)
        s *= 2
    return s


SLICE_ALIGN = 64           # first-level slices: a multiple of 64 rows (64 x 16320 rows ran like 64 x 16128: 552 vs 551 us, and leave 4095 instead of 16383 rows)
TAIL_SLICE = 256           # second level: the remainder in 256-row slices
_MASKED_TAIL_ON_HOST = False   # tests only: take _tail_product's masked-slice branch (the GPU's) for host tensors too, so its indexing is covered without a GPU


def split_plan(rows, out_elems=None):
    """How a contraction over ``rows`` positions is cut into batched slices: ([(first row, slices, rows per slice), ...], rows covered).
    ``rows`` divisible by split_count(rows): one level, as in rounds 1 - 4.  Otherwise -- the reference trainer's B L = B (max_length - 1) --
    the slices of the first level are shortened to a multiple of 64 rows (the library's kernels for an ODD contraction length ran 12 - 17 %
    slower -- 64 x 16383 rows 645 us, 64 x 16320 rows 552 us for in_proj's gradient -- and the rows % S leftover went through an fp32 product
    behind two conversion passes), a second level takes the remainder in 256-row slices, and what is left (< 256 rows) goes through one small
    fp32 product."""
    s = split_count(rows, out_elems)
    q = rows // s
    # (round 6: equal slices of an ODD number of rows are the slow case themselves -- 10^6 rows = 64 x 15625: dW_in 549 vs 441 us, dW1 693 vs 583 with the
    # two-level plan, profiles/r6_wgrad_plan.txt -- so they are cut like a row count the slice count does not divide)
    if (rows % s == 0 and q % 2 == 0) or q < 2 * TAIL_SLICE:
        return [(0, s, q)], s * q
    q -= q % SLICE_ALIGN
    levels, pos = [(0, s, q)], s * q
    s2 = (rows - pos) // TAIL_SLICE
    if s2 > 0:
        levels.append((pos, s2, TAIL_SLICE))
        pos += s2 * TAIL_SLICE
    return levels, pos


def _tail_product(a_cn, b_nk, done):
    """sum over the positions p >= done of a_cn[:, p] b_nk[p, :] in fp32 -- the < 256 rows split_plan leaves -- for views a_cn (C, n), b_nk (n, K) of
    any strides.  On the GPU: the LAST 256 positions as one more 16-bit slice whose already-counted columns are zeroed in a small copy (an fp32
    product of 255 rows behind two conversion passes took 80 + 12 us: the library schedules it on a handful of workgroups)."""
    n = a_cn.shape[1]
    r = n - done
    if r <= 0:
        return None
    if not (a_cn.is_cuda or _MASKED_TAIL_ON_HOST) or n < TAIL_SLICE or a_cn.dtype == torch.float32:
        return torch.mm(a_cn[:, done:].float(), b_nk[done:].float())
    a = a_cn[:, n - TAIL_SLICE:].clone()
    a[:, :TAIL_SLICE - r] = 0
    return _bmm_f32(a.unsqueeze(0), b_nk[n - TAIL_SLICE:].unsqueeze(0))[0]


# What split_plan leaves behind its first level -- the 256-row second level and the tail, up to ~4160 rows -- as ONE batched product over zero-padded
# copies (round 6: the two small products, their partial sums and additions cost a big weight gradient 65 - 100 us at 2^20 - 1 rows against ~45 this way,
# profiles/r6_wgrad_plan.txt).  HYENA_WGRAD_LEFTOVER=split: the separate second level + masked tail of round 5 (A/B).
MERGE_LEFTOVER = os.environ.get("HYENA_WGRAD_LEFTOVER", "merged") != "split"


def _leftover_product(a_cn, b_nk, start):
    """sum over the positions p >= start of a_cn[:, p] b_nk[p, :] in fp32, for views a_cn (C, n), b_nk (n, K) of any strides: the LAST w positions, w the
    leftover rounded up to whole 256-row slices, as one batched product -- a_cn's slice copied with its already-counted leading columns zeroed (what
    _tail_product does for one slice), b_nk's taken as it lies (its leading rows meet zeros); the slices' sums added in order"""
    n = a_cn.shape[1]
    r = n - start
    if r <= 0:
        return None
    s2 = (r + TAIL_SLICE - 1) // TAIL_SLICE
    w = s2 * TAIL_SLICE
    if not (a_cn.is_cuda or _MASKED_TAIL_ON_HOST) or a_cn.dtype == torch.float32 or n < w:
        return torch.mm(a_cn[:, start:].float(), b_nk[start:].float())
    A = a_cn[:, n - w:].clone(memory_format=torch.contiguous_format)
    if w > r:
        A[:, :w - r].zero_()
    Bv = b_nk[n - w:]
    return _bmm_f32(A.view(a_cn.shape[0], s2, TAIL_SLICE).permute(1, 0, 2), Bv.reshape(s2, TAIL_SLICE, b_nk.shape[1])).sum(0)


MERGE_MAX_ROWS = 4352          # a 256 x 256 gradient is cut into 256 first-level slices and leaves up to 16383 rows: copying those costs more than the
                               # second product saves (dW_out at 2^20 - 1: 303 -> 333 us merged) -- such leftovers keep round 5's view-based second level


def _use_merged(t, levels=None, rows=0):
    if not (MERGE_LEFTOVER and (t.is_cuda or _MASKED_TAIL_ON_HOST)):
        return False
    return levels is None or rows - (levels[0][0] + levels[0][1] * levels[0][2]) <= MERGE_MAX_ROWS


def split_k_weight_grad(dy2, x2):
    """dy2^T x2 (fp32) as batched position slices (split_plan) + the leftover rows, partial sums added in a fixed order (deterministic)."""
    rows, n = dy2.shape
    k = x2.shape[1]
    levels, done = split_plan(rows, n * k)
    merged = _use_merged(dy2, levels, rows)
    if merged:
        levels = levels[:1]
        done = levels[0][0] + levels[0][1] * levels[0][2]
    dw = None
    for p0, s, q in levels:
        g = _bmm_f32(dy2[p0:p0 + s * q].view(s, q, n).transpose(1, 2), x2[p0:p0 + s * q].view(s, q, k)).sum(0)
        dw = g if dw is None else dw + g
    if done < rows:                                                 # the leftover rows
        dw = dw + (_leftover_product(dy2.t(), x2, done) if merged else _tail_product(dy2.t(), x2, done))
    return dw


class SplitKLinearFunc(torch.autograd.Function):
    """``dt``: the 16-bit compute type under autocast -- ``weight`` / ``bias`` are then the fp32 PARAMETERS, used through their per-step shadows
    (_castcache), and their gradients go back in the parameters' own type; None: the operands as they come"""

    @staticmethod
    def forward(ctx, x, weight, bias, dt=None):
        w = weight if dt is None else _castcache.shadow(weight, dt)
        b = bias if dt is None or bias is None else _castcache.shadow(bias, dt)
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.ptypes = (weight.dtype, None if bias is None else bias.dtype, dt)
        return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        wdt, bdt, dt = ctx.ptypes
        n, k = weight.shape
        dy2 = dy.reshape(-1, n)
        x2 = x.reshape(-1, k)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.mm(dy2, weight).view(x.shape)
        if ctx.needs_input_grad[1]:
            dw = _castcache.wgrad_out(split_k_weight_grad(dy2, x2), wdt, dt)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            from . import _lib
            db = _lib.colsum(dy2.contiguous()) if dy2.is_cuda else dy2.sum(0, dtype=torch.float32)
            db = _castcache.wgrad_out(db, bdt, dt if dt is not None else dy.dtype)
        return dx, dw, db, None


def hyena_linear(x, weight, bias):
    """``F.linear(x, weight, bias)`` with the autocast semantics of ``nn.Linear`` and the split-K weight gradient."""
    if x.is_cuda and x.shape[-1] == weight.shape[1] and x.numel() // x.shape[-1] >= MIN_ROWS:
        if torch.is_autocast_enabled():
            dt = torch.get_autocast_dtype("cuda")
            if dt in (torch.bfloat16, torch.float16):
                with torch.autocast("cuda", enabled=False):
                    return SplitKLinearFunc.apply(x.to(dt).contiguous(), weight, bias, dt)
        elif x.dtype == weight.dtype and x.dtype in (torch.float32, torch.bfloat16, torch.float16):
            return SplitKLinearFunc.apply(x.contiguous(), weight, bias)          # plain fp32 / 16-bit training: same pathology
    return F.linear(x, weight, bias)


# ---------------------------------------------------------------------------------------------------------------------
# Channel-major projections (what hyena_dna_amd.mixer's channel-major core sits between).  The GEMM library takes transposed
# operands at no cost (profiles/gemm_layout_r2.txt), so in_proj can WRITE x^T = W u^T (3D, B L) and out_proj can READ z^T
# (D, B L): the two rearranging copies of hyena.py:392 and hyena.py:432-439 are never made.
# ---------------------------------------------------------------------------------------------------------------------
def _bmm_f32(a, b):
    if a.is_cuda:
        return torch.bmm(a, b, out_dtype=torch.float32)
    return torch.bmm(a.float(), b.float())


# Channel-major tensors have PITCHED channel rows since round 5 (hyena_dna_amd._lib.empty_cm: the reference trainer's L = max_length - 1 is odd,
# and a packed (C, B, L) tensor then has no aligned row but the first): row (c, b) at c cs + b L with cs = B L rounded up to 64 elements.  A
# library GEMM takes such an operand as ONE (C, B L) matrix with leading dimension cs (_lib.cm_matrix) -- the products and their summation
# order are the ones of rounds 2 - 4.  A tensor whose sequences are pitched individually (bs > L) has no such view: one product per sequence.
def _pieces(t, L):
    """t (C, B, L) channel-major -> [(C, n) matrix view, first flattened position, n]: one piece if the sequences of a channel are adjacent"""
    from . import _lib
    C, B, _ = t.shape
    m = _lib.cm_matrix(t)
    if m is not None:
        return [(m, 0, B * L)]
    return [(t[:, b, :], b * L, L) for b in range(B)]


def cm_from_pm(w, x2, B, L):
    """(C, B, L) channel-major (_lib.empty_cm layout) = w (C, K) x2^T for position-major x2 (B L, K)"""
    from . import _lib
    out = _lib.empty_cm(w.shape[0], B, L, x2.dtype, x2.device)
    for m, p0, n in _pieces(out, L):
        torch.mm(w, x2[p0:p0 + n].t(), out=m)
    return out


def pm_from_cm(t, w, bias=None):
    """(B L, N) position-major = t^T w (+ bias) for t (C, B, L) packed or pitched rows, w (C, N)"""
    C, B, L = t.shape
    pieces = _pieces(t, L)
    if len(pieces) == 1:
        m = pieces[0][0]
        return torch.mm(m.t(), w) if bias is None else torch.addmm(bias, m.t(), w)
    out = torch.empty((B * L, w.shape[1]), dtype=t.dtype, device=t.device)
    for m, p0, n in pieces:
        if bias is None:
            torch.mm(m.t(), w, out=out[p0:p0 + n])
        else:
            torch.addmm(bias, m.t(), w, out=out[p0:p0 + n])
    return out


def wgrad_cm_pm(d, x2):
    """sum over the positions of d[c, p] x2[p, k] -> (C, K) fp32 for d (C, B, L) packed or pitched rows and position-major x2 (B L, K): S batched
    position slices + a tail per piece, partial sums added in a fixed order (deterministic)"""
    C, B, L = d.shape
    k = x2.shape[1]
    total = None
    for m, p0, n in _pieces(d, L):
        xs = x2[p0:p0 + n]
        levels, done = split_plan(n, C * k)
        merged = _use_merged(d, levels, n)
        if merged:
            levels = levels[:1]
            done = levels[0][0] + levels[0][1] * levels[0][2]
        for r0, s, q in levels:
            g = _bmm_f32(m[:, r0:r0 + s * q].reshape(C, s, q).permute(1, 0, 2), xs[r0:r0 + s * q].view(s, q, k)).sum(0)
            total = g if total is None else total + g
        if done < n:
            total = total + (_leftover_product(m, xs, done) if merged else _tail_product(m, xs, done))
    return total


def wgrad_pm_cm(dy2, z):
    """sum over the positions of dy2[p, n] z[k, p] -> (N, K) fp32 for position-major dy2 (B L, N) and z (K, B, L) packed or pitched rows"""
    K, B, L = z.shape
    N = dy2.shape[1]
    total = None
    for m, p0, n in _pieces(z, L):
        ds = dy2[p0:p0 + n]
        levels, done = split_plan(n, N * K)
        merged = _use_merged(z, levels, n)
        if merged:
            levels = levels[:1]
            done = levels[0][0] + levels[0][1] * levels[0][2]
        for r0, s, q in levels:
            g = _bmm_f32(ds[r0:r0 + s * q].view(s, q, N).transpose(1, 2), m[:, r0:r0 + s * q].reshape(K, s, q).permute(1, 2, 0)).sum(0)
            total = g if total is None else total + g
        if done < n:
            total = total + (_leftover_product(ds.t(), m.t(), done) if merged else _tail_product(ds.t(), m.t(), done))
    return total


class InProjCMFunc(torch.autograd.Function):
    """xT (N, B, L) = W (N, K) u^T, WITHOUT the bias (the shell kernels add it on load and return its gradient)."""

    @staticmethod
    def forward(ctx, u, weight, dt=None):
        B, L, K = u.shape
        u2 = u.reshape(B * L, K)
        w = weight if dt is None else _castcache.shadow(weight, dt)           # (dt: autocast's compute type; weight is then the fp32 parameter)
        ctx.save_for_backward(u2, w)
        ctx.ushape = u.shape
        ctx.ptypes = (weight.dtype, dt)
        return cm_from_pm(w, u2, B, L)

    @staticmethod
    def backward(ctx, dxT):
        from . import _lib
        u2, weight = ctx.saved_tensors
        dxT = _lib.as_cm(dxT)
        du = dw = None
        if ctx.needs_input_grad[0]:
            du = pm_from_cm(dxT, weight).view(ctx.ushape)
        if ctx.needs_input_grad[1]:
            dw = _castcache.wgrad_out(wgrad_cm_pm(dxT, u2), *ctx.ptypes)
        return du, dw, None


class InProjPreCMFunc(torch.autograd.Function):
    """xT (N, B, L) = W u^T as InProjCMFunc, from this package's matrix-core kernel (csrc/proj_kernels.h), which also hands back
    vg = short_conv(xT + b_in)[v] * short_conv(xT + b_in)[x1] -- the long convolution's input -- from its epilogue.  vg is a cached
    value for HyenaMixerCMFunc (whose backward recomputes it from xT), not a differentiable output; the gradients of the projection
    are the library GEMMs of InProjCMFunc."""

    @staticmethod
    def forward(ctx, u, weight, b_in, sf_weight, sf_bias, L, pad_to=0, dt=None):
        from . import _lib
        B, Lx, K = u.shape
        ctx.ptypes = (weight.dtype, dt)
        if dt is not None:
            weight = _castcache.shadow(weight, dt)                           # (dt: autocast's compute type; weight is then the fp32 parameter)
        ctx.narrow = None
        if pad_to > Lx:
            # several sequences of a length that is not a multiple of 64 (the reference trainer's L = max_length - 1): the kernels run on sequences
            # padded with zero positions to the next multiple -- xT / dxT rows then start aligned, as at the neighbouring aligned length, for one
            # extra pass over u (off by default: measured slower, see PAD_SEQUENCES below).
            # Zero positions give zero rows of xT (it carries no bias), lie beyond the convolved length L and receive zero gradients.
            up = torch.empty((B, pad_to, K), dtype=u.dtype, device=u.device)
            up[:, :Lx].copy_(u)
            up[:, Lx:].zero_()
            ctx.narrow = Lx
            u, Lx = up, pad_to
        ctx.save_for_backward(u.reshape(B * Lx, K), weight)
        ctx.ushape = u.shape
        bi = None if b_in is None else b_in.detach().to(torch.float32).contiguous()
        w = sf_weight.detach().to(torch.float32).reshape(weight.shape[0], 3).contiguous()
        b = sf_bias.detach().to(torch.float32).contiguous()
        xT, vg = _lib.inproj_pre_fwd(u, weight, bi, w, b, L)
        ctx.mark_non_differentiable(vg)
        # (otherwise autograd hands backward a zero tensor for vg's "gradient": a 512 MB fill per layer at L = 2^20)
        ctx.set_materialize_grads(False)
        return xT, vg

    @staticmethod
    def backward(ctx, dxT, _dvg):
        if dxT is None:
            return None, None, None, None, None, None, None, None
        du, dw, _ = InProjCMFunc.backward(ctx, dxT)
        if du is not None and ctx.narrow is not None:
            du = du[:, :ctx.narrow]
        return du, dw, None, None, None, None, None, None


INPROJ_MFMA = os.environ.get("HYENA_INPROJ_MFMA", "1") != "0"      # A/B knob: 0 = library GEMM + cm_pre_fwd


# Several sequences of a length that is not a multiple of 64 on zero-padded positions inside in_proj (InProjPreCMFunc, pad_to): built in round 5,
# values identical, MEASURED SLOWER and therefore off -- 32767 x 8: the layer 2.528 vs 2.496 ms, 159999 x 2: 3.894 vs 3.814 ms (aligned neighbours
# 2.318 / 3.740; profiles/r5_stress.txt): it aligns xT / dxT only -- zT, dzT and the position-major side keep B (max_length - 1) positions -- and costs a
# pass over u.  HYENA_PAD_SEQUENCES=1 turns it on.
PAD_SEQUENCES = os.environ.get("HYENA_PAD_SEQUENCES", "0") == "1"


def in_proj_pre_cm(u, weight, b_in, sf_weight, sf_bias, L):
    """(B, Lx, K) -> (xT (N, B, Lx), vg (B, D, L) or None).  Where the matrix-core kernel serves the call (16-bit operands --
    autocast or plain --, d_model 128 / 256, short_filter_order 3) both come from ONE launch; otherwise xT is the library GEMM of
    in_proj_cm and vg is None (hyena_mixer_core_cm then runs cm_pre_fwd)."""
    from . import _lib
    dt = _autocast_dtype(u) or (u.dtype if u.dtype in (torch.bfloat16, torch.float16) and weight.dtype == u.dtype else None)
    B, Lx, K = u.shape
    if (INPROJ_MFMA and dt is not None and weight.shape == (3 * K, K) and sf_weight.shape[-1] == 3 and sf_weight.shape[0] == 3 * K
            and (u.is_cuda or _lib._backend.name != "hip") and L >= 1 and _lib.proj_supported(B, Lx, K, dt)):
        pad_to = _lib.row_pitch(Lx) if (PAD_SEQUENCES and B > 1 and Lx == L and Lx % _lib.ROW_ALIGN != 0 and Lx >= 4 * _lib.ROW_ALIGN) else 0
        with torch.autocast("cuda" if u.is_cuda else "cpu", enabled=False):
            if weight.dtype == torch.float32:          # (autocast over fp32 parameters: the weight through its per-step shadow)
                return InProjPreCMFunc.apply(u.to(dt).contiguous(), weight, b_in, sf_weight, sf_bias, L, pad_to, dt)
            return InProjPreCMFunc.apply(u.to(dt).contiguous(), weight.to(dt).contiguous(), b_in, sf_weight, sf_bias, L, pad_to)
    return in_proj_cm(u, weight), None


class OutProjCMFunc(torch.autograd.Function):
    """y (B, L, N) = zT^T W^T + b for zT (K, B, L)."""

    @staticmethod
    def forward(ctx, zT, weight, bias, dt=None):
        from . import _lib
        K, B, L = zT.shape
        zT = _lib.as_cm(zT)
        ctx.ptypes = (weight.dtype, None if bias is None else bias.dtype, dt)
        if dt is not None:                                                   # (autocast's compute type; weight / bias are then the fp32 parameters)
            weight, bias = _castcache.shadow(weight, dt), _castcache.shadow(bias, dt)
        ctx.save_for_backward(zT, weight)
        ctx.has_bias = bias is not None
        return pm_from_cm(zT, weight.t(), bias).view(B, L, weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        zT, weight = ctx.saved_tensors
        n, k = weight.shape
        K, B, L = zT.shape
        rows = B * L
        dy2 = dy.reshape(rows, n)
        dz = dw = db = None
        if ctx.needs_input_grad[0]:
            dz = cm_from_pm(weight.t(), dy2, B, L)                            # (K, B, L): channel-major, straight from the GEMM
        wdt, bdt, dt = ctx.ptypes
        if ctx.needs_input_grad[1]:
            dw = _castcache.wgrad_out(wgrad_pm_cm(dy2, zT), wdt, dt)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            from . import _lib
            db = _castcache.wgrad_out(_lib.colsum(dy2.contiguous()), bdt, dt if dt is not None else dy.dtype)
        return dz, dw, db, None


def _autocast_dtype(x):
    if torch.is_autocast_enabled():
        dt = torch.get_autocast_dtype("cuda" if x.is_cuda else "cpu")
        if dt in (torch.bfloat16, torch.float16):
            return dt
    return None


def in_proj_cm(u, weight):
    """(B, L, K) -> xT (N, B, L) = W u^T with nn.Linear's autocast semantics (bias NOT added: see InProjCMFunc)."""
    dt = _autocast_dtype(u)
    if dt is not None:
        with torch.autocast("cuda" if u.is_cuda else "cpu", enabled=False):
            if weight.dtype == torch.float32:
                return InProjCMFunc.apply(u.to(dt).contiguous(), weight, dt)
            return InProjCMFunc.apply(u.to(dt).contiguous(), weight.to(dt))
    return InProjCMFunc.apply(u.contiguous(), weight.to(u.dtype))


def out_proj_cm(zT, weight, bias):
    """zT (K, B, L) -> (B, L, N) = zT^T W^T + b with nn.Linear's autocast semantics."""
    dt = _autocast_dtype(zT)
    if dt is not None:
        with torch.autocast("cuda" if zT.is_cuda else "cpu", enabled=False):
            if weight.dtype == torch.float32 and (bias is None or bias.dtype == torch.float32):
                return OutProjCMFunc.apply(zT.to(dt), weight, bias, dt)
            return OutProjCMFunc.apply(zT.to(dt), weight.to(dt), None if bias is None else bias.to(dt))
    return OutProjCMFunc.apply(zT, weight.to(zT.dtype), None if bias is None else bias.to(zT.dtype))
