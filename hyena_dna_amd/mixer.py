"""The fused core of ``HyenaOperator.forward`` between the two projections (reference:
``src/models/sequence/hyena.py:392-439``): short depthwise conv (k = 3), ``v * x1`` gate, long convolution,
``* x0`` gate and the (B, L, C) <-> (B, D, L) layout changes, as one autograd function over the HIP kernels of
``include/hyena_mixer.h`` + ``include/hyena_fftconv.h``.

Valid for the HyenaDNA operator configuration: order 2, one head, one block, inner factor 1, no outer mixing /
post-order FFN, dropout 0, identity activation, ``short_filter_order`` 3.  ``hyena_dna_amd.hyena.HyenaOperator`` uses it
whenever those hold and takes its generic (PyTorch-glue + ``fftconv_func``) path otherwise.

What is kept for the backward: ``x`` (the in_proj output, alive in autograd anyway), ``y`` (the conv output), the filter
and -- unless disabled -- the forward's column spectra; the three short-conv outputs are recomputed from ``x`` inside the
backward kernels (3 taps) instead of being stored.
"""
import torch

from . import _castcache, _gradmode, _lib

__all__ = ["hyena_mixer_core", "HyenaMixerFunc", "hyena_mixer_core_cm", "HyenaMixerCMFunc", "hyena_mixer_out_cm", "HyenaMixerOutCMFunc",
           "mixer_out_supported", "hyena_mixer_core_cm_order_n", "HyenaMixerCMOrderNFunc"]


class HyenaMixerFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, sf_weight, sf_bias, k, bias, L):
        """x (B, Lx, 3D); sf_weight (3D, 1, 3); sf_bias (3D,); k (D, L) fp32; bias (D,) -> z (B, L, D), L <= Lx."""
        B, Lx, D3 = x.shape
        D = D3 // 3
        xc = x.contiguous()
        w = sf_weight.detach().to(torch.float32).reshape(D3, 3).contiguous()
        b = sf_bias.detach().to(torch.float32).contiguous()
        kf = k.detach().to(torch.float32).contiguous()
        bf = bias.detach().to(torch.float32).reshape(D).contiguous()
        vg = _lib.mixer_pre_fwd(xc, w, b, L)
        want_grad = any(_gradmode.needs(ctx)[:5])
        spectra = None
        if want_grad and _lib.save_spectra_default(B, D, L, device=vg.device):
            y, spectra = _lib.fftconv_fwd(vg, kf, bf, save=True)
        else:
            y = _lib.fftconv_fwd(vg, kf, bf, grad=want_grad)
        z = _lib.mixer_post_fwd(y, xc, w, b)
        ctx.save_for_backward(xc, w, b, kf, bf, y)
        ctx.spectra = spectra
        ctx.meta = (sf_weight.shape, sf_weight.dtype, sf_bias.dtype, k.dtype, bias.shape, bias.dtype, L)
        return z

    @staticmethod
    def backward(ctx, dz):
        xc, w, b, kf, bf, y = ctx.saved_tensors
        w_shape, w_dtype, b_dtype, k_dtype, bias_shape, bias_dtype, L = ctx.meta
        B, Lx, D3 = xc.shape
        dz = dz.to(xc.dtype).contiguous()
        dx = torch.zeros_like(xc) if Lx > L else torch.empty_like(xc)
        part = _lib.mixer_partials(xc, L)
        dy = _lib.mixer_post_bwd(dz, y, xc, w, b, dx, part)
        # the conv's input v * x1 is recomputed from x -- unless the saved spectra hold its transform (two-level plan)
        need_vg = ctx.spectra is None or _lib.lib().hyena_fftconv_plan(int(L)) == _lib.PLAN_ONCHIP
        vg = _lib.mixer_pre_fwd(xc, w, b, L) if need_vg else None
        need_dk = ctx.needs_input_grad[3] or ctx.needs_input_grad[4]          # a frozen filter skips the dk path entirely
        dvg, dk, dbias = _lib.fftconv_bwd(dy, vg, kf, bf, need_du=True, need_dk=need_dk, saved=ctx.spectra)
        ctx.spectra = None
        _lib.mixer_pre_bwd(dvg, xc, w, b, dx, part)
        red = part.sum(dim=(0, 1))                                  # (3D, 4): deterministic two-stage reduction
        dw = red[:, :3].reshape(w_shape).to(w_dtype)
        db = red[:, 3].to(b_dtype)
        return (dx, dw, db, dk.to(k_dtype) if dk is not None else None,
                dbias.reshape(bias_shape).to(bias_dtype) if dbias is not None else None, None)


def hyena_mixer_core(x, sf_weight, sf_bias, k, bias, L):
    """z = ((fftconv(v * x1, k, bias)) * x0)^T with (x0, x1, v) = short_conv(x^T)[..., :L].split(D)."""
    if x.shape[0] == 0 or L == 0:
        # empty batch (or length): nothing to launch.  Like the reference's PyTorch ops this returns an empty tensor that
        # is still connected to every input (their gradients are zeros, not None).
        D = x.shape[-1] // 3
        zero = 0 * (sf_weight.sum() + sf_bias.sum() + k.sum() + bias.sum())
        return x[:, :L, :D] * 0 + zero.to(x.dtype)
    return _gradmode.apply(HyenaMixerFunc, x, sf_weight, sf_bias, k, bias, L)


class HyenaMixerCMFunc(torch.autograd.Function):
    """The same core in channel-major layout (``csrc/cm_kernels.h``): ``xT`` (3D, B, Lx) = W_in u^T without the in_proj bias,
    result ``zT`` (D, B, L) for ``projection.out_proj_cm``.  No tensor between the two projections is ever transposed."""

    @staticmethod
    def forward(ctx, xT, b_in, sf_weight, sf_bias, k, bias, L, vg=None):
        # vg: the conv's input v * x1 if the projection kernel already produced it (projection.in_proj_pre_cm: bit-identical to
        # cm_pre_fwd on this xT) -- a cached value, not a differentiable input
        D3, B, Lx = xT.shape
        D = D3 // 3
        xc = _lib.as_cm(xT)
        bi = b_in.detach().to(torch.float32).contiguous()
        w = sf_weight.detach().to(torch.float32).reshape(D3, 3).contiguous()
        b = sf_bias.detach().to(torch.float32).contiguous()
        kf = _lib.as_rows(k.detach().to(torch.float32))
        bf = bias.detach().to(torch.float32).reshape(D).contiguous()
        if vg is None:
            vg = _lib.cm_pre_fwd(xc, bi, w, b, L)
        want_grad = any(_gradmode.needs(ctx)[:6])
        spectra = None
        if want_grad and _lib.save_spectra_default(B, D, L, device=vg.device):
            y, spectra = _lib.fftconv_fwd(vg, kf, bf, save=True)
        else:
            y = _lib.fftconv_fwd(vg, kf, bf, grad=want_grad)
        zT = _lib.cm_post_fwd(y, xc, bi, w, b)
        ctx.save_for_backward(xc, bi, w, b, kf, bf, y)
        ctx.spectra = spectra
        ctx.meta = (b_in.dtype, sf_weight.shape, sf_weight.dtype, sf_bias.dtype, k.dtype, bias.shape, bias.dtype, L)
        return zT

    @staticmethod
    def backward(ctx, dzT):
        xc, bi, w, b, kf, bf, y = ctx.saved_tensors
        bin_dtype, w_shape, w_dtype, b_dtype, k_dtype, bias_shape, bias_dtype, L = ctx.meta
        D3, B, Lx = xc.shape
        dzT = _lib.as_cm(dzT.to(xc.dtype))
        dxT = _lib.empty_like_cm(xc)
        if Lx > L:
            dxT[:, :, L:].zero_()                     # (the kernels write every position < L of every row)
        part = _lib.cm_partials(xc, L)
        dy = _lib.cm_post_bwd(dzT, y, xc, bi, w, b, dxT, part)
        need_vg = ctx.spectra is None or _lib.lib().hyena_fftconv_plan(int(L)) == _lib.PLAN_ONCHIP
        vg = _lib.cm_pre_fwd(xc, bi, w, b, L) if need_vg else None                 # recompute the conv's input
        need_dk = ctx.needs_input_grad[4] or ctx.needs_input_grad[5]
        dvg, dk, dbias = _lib.fftconv_bwd(dy, vg, kf, bf, need_du=True, need_dk=need_dk, saved=ctx.spectra)
        ctx.spectra = None
        _lib.cm_pre_bwd(dvg, xc, bi, w, b, dxT, part)
        red = part[:, :, :5].sum(dim=1)                              # (3D, 5): deterministic two-stage reduction
        dw = red[:, :3].reshape(w_shape).to(w_dtype)
        db = red[:, 3].to(b_dtype)
        dbin = red[:, 4].to(bin_dtype)
        return (dxT, dbin, dw, db, dk.to(k_dtype) if dk is not None else None,
                dbias.reshape(bias_shape).to(bias_dtype) if dbias is not None else None, None, None)


def hyena_mixer_core_cm(xT, b_in, sf_weight, sf_bias, k, bias, L, vg=None):
    """zT (D, B, L) = fftconv(v * x1, k, bias) * x0 with (x0, x1, v) = short_conv(xT + b_in)[..., :L].split(D) (channel-major)."""
    if xT.shape[1] == 0 or L == 0:
        D = xT.shape[0] // 3
        zero = 0 * (b_in.sum() + sf_weight.sum() + sf_bias.sum() + k.sum() + bias.sum())
        return xT[:D, :, :L] * 0 + zero.to(xT.dtype)
    return _gradmode.apply(HyenaMixerCMFunc, xT, b_in, sf_weight, sf_bias, k, bias, L, vg)


# ---------------------------------------------------------------------------------------------------------------------
# Order >= 3 (round 6; configs/model/layer/hyena_dna.yaml:3 ships ``order: 3``).  hyena.py:404-439 with n = order - 1 long convolutions:
#     (x_0 ... x_n, v) = short_conv(x^T + b_in).split(D);   v <- conv(v * x_n, k_0);   v <- conv(v * x_{n-o}, k_o)  (o = 1 ... n - 1);   z = v * x_0
# No new kernel: every gate is one of the order-2 shell kernels on a THREE-GROUP ROW VIEW of x^T (channel rows lead, so x^T[s D : (s + 3) D] is a
# channel-major tensor of its own) --
#     v * x_n            = cm_pre_fwd  on the view that starts at group n - 1   (its groups 1, 2 are x_n, v)
#     y_{o-1} * x_{n-o}  = cm_post_fwd on the view that starts at group n - o   (its group 0), written as the next convolution's (B, D, L) rows
#     y_{n-1} * x_0      = cm_post_fwd on the view that starts at group 0, channel-major for out_proj
# and the backward runs the same views in reverse: every group of dx^T is written exactly once (group 0 and groups 1 ... n - 1 by cm_post_bwd,
# groups n, n + 1 by cm_pre_bwd), the short-filter / in_proj-bias records of each launch cover the groups it wrote.
# ---------------------------------------------------------------------------------------------------------------------
class HyenaMixerCMOrderNFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xT, b_in, sf_weight, sf_bias, bias, L, order, *ks):
        """xT ((order + 1) D, B, Lx) = W_in u^T without the bias; ks: order - 1 filters (D, L), one per convolution (HyenaFilter.filter_dl_split);
        bias (D (order - 1),) in '(v o)' order (hyena.py:410-412) -> zT (D, B, L)"""
        G = order + 1
        n = order - 1
        GD, B, Lx = xT.shape
        D = GD // G
        xc = _lib.as_cm(xT)
        bi = b_in.detach().to(torch.float32).contiguous()
        w = sf_weight.detach().to(torch.float32).reshape(GD, 3).contiguous()
        b = sf_bias.detach().to(torch.float32).contiguous()
        kfs = []
        for k in ks:
            kf = k.detach().to(torch.float32)
            ld = _lib.ld_of(kf)
            if ld is None or ld > _lib.row_pitch(L):               # a row view of a '(v o)' block: gather it (dk comes back with k's pitch)
                rows = _lib.empty_rows((D,), L, torch.float32, kf.device)
                rows.copy_(kf)
                kf = rows
            kfs.append(_lib.as_rows(kf))
        bf = bias.detach().to(torch.float32).reshape(D, n).t().contiguous()          # (n, D)
        want_grad = any(_gradmode.needs(ctx))

        def view(s):
            return xc[s * D:(s + 3) * D], bi[s * D:(s + 3) * D], w[s * D:(s + 3) * D], b[s * D:(s + 3) * D]

        v = _lib.cm_pre_fwd(*view(n - 1), L)
        ys = []
        spectra = [None] * n
        keep = want_grad and _lib.save_spectra_default(B, D, L, device=v.device)      # the forward's spectra for the backward, as HyenaMixerCMFunc keeps them
        for o in range(n):
            if keep:
                y, spectra[o] = _lib.fftconv_fwd(v, kfs[o], bf[o], save=True)
            else:
                y = _lib.fftconv_fwd(v, kfs[o], bf[o], grad=want_grad)
            ys.append(y)
            if o + 1 < n:
                v = _lib.cm_post_fwd(y, *view(n - o - 1), rows_out=True)
        zT = _lib.cm_post_fwd(ys[-1], *view(0))
        ctx.save_for_backward(xc, bi, w, b, bf, *kfs, *ys)
        ctx.spectra = spectra
        ctx.meta = (b_in.dtype, sf_weight.shape, sf_weight.dtype, sf_bias.dtype, [k.dtype for k in ks], bias.shape, bias.dtype, L, order)
        return zT

    @staticmethod
    def backward(ctx, dzT):
        xc, bi, w, b, bf, *rest = ctx.saved_tensors
        bin_dtype, w_shape, w_dtype, b_dtype, k_dtypes, bias_shape, bias_dtype, L, order = ctx.meta
        G, n = order + 1, order - 1
        kfs, ys = rest[:n], rest[n:]
        GD, B, Lx = xc.shape
        D = GD // G
        dzT = _lib.as_cm(dzT.to(xc.dtype))
        dxT = _lib.empty_like_cm(xc)
        if Lx > L:
            dxT[:, :, L:].zero_()

        def view(s):
            return xc[s * D:(s + 3) * D], bi[s * D:(s + 3) * D], w[s * D:(s + 3) * D], b[s * D:(s + 3) * D]

        need_dk = ctx.needs_input_grad[4] or any(ctx.needs_input_grad[7:])
        onchip = _lib.lib().hyena_fftconv_plan(int(L)) == _lib.PLAN_ONCHIP
        red = torch.empty(GD, 5, dtype=torch.float32, device=xc.device)
        # z = y_{n-1} * x_0
        part = _lib.cm_partials(view(0)[0], L)
        dy = _lib.cm_post_bwd(dzT, ys[n - 1], *view(0), dxT[0:3 * D], part)
        red[0:D] = part[:D, :, :5].sum(dim=1)
        dks, dbs = [None] * n, [None] * n
        for o in range(n - 1, -1, -1):
            # the input of convolution o, recomputed (one elementwise pass) instead of kept -- unless the saved spectra hold its transform (two-level plan)
            sp = ctx.spectra[o]
            ctx.spectra[o] = None
            if sp is not None and not onchip:
                v = None
            elif o == 0:
                v = _lib.cm_pre_fwd(*view(n - 1), L)
            else:
                v = _lib.cm_post_fwd(ys[o - 1], *view(n - o), rows_out=True)
            dv, dks[o], dbs[o] = _lib.fftconv_bwd(dy, v, kfs[o], bf[o], need_du=True, need_dk=need_dk, saved=sp)
            del sp
            s = n - o if o > 0 else n - 1
            part = _lib.cm_partials(view(s)[0], L)
            if o > 0:                                                 # v = y_{o-1} * x_{n-o}: group 0 of the view at n - o
                dy = _lib.cm_post_bwd(dv, ys[o - 1], *view(s), dxT[s * D:(s + 3) * D], part, dz_rows=True)
                red[s * D:(s + 1) * D] = part[:D, :, :5].sum(dim=1)
            else:                                                     # v = x_n * v: groups 1, 2 of the view at n - 1
                _lib.cm_pre_bwd(dv, *view(s), dxT[s * D:(s + 3) * D], part)
                red[(s + 1) * D:(s + 3) * D] = part[D:, :, :5].sum(dim=1)
        dw = red[:, :3].reshape(w_shape).to(w_dtype)
        db = red[:, 3].to(b_dtype)
        dbin = red[:, 4].to(bin_dtype)
        dbias = None
        if need_dk:
            dbias = torch.stack(dbs, dim=1).reshape(bias_shape).to(bias_dtype)            # back to '(v o)' order
            dks = [dk.to(t) for dk, t in zip(dks, k_dtypes)]
        return (dxT, dbin, dw, db, dbias, None, None, *dks)


def hyena_mixer_core_cm_order_n(xT, b_in, sf_weight, sf_bias, ks, bias, L, order):
    """zT (D, B, L) of an operator of order >= 2 from xT ((order + 1) D, B, Lx), channel-major; ks: its order - 1 filters (HyenaMixerCMOrderNFunc)"""
    assert len(ks) == order - 1
    if xT.shape[1] == 0 or L == 0:
        D = xT.shape[0] // (order + 1)
        zero = 0 * (b_in.sum() + sf_weight.sum() + sf_bias.sum() + sum(k.sum() for k in ks) + bias.sum())
        return xT[:D, :, :L] * 0 + zero.to(xT.dtype)
    return _gradmode.apply(HyenaMixerCMOrderNFunc, xT, b_in, sf_weight, sf_bias, bias, L, order, *ks)


# ---------------------------------------------------------------------------------------------------------------------
# The channel-major core WITH out_proj (round 4): the second gate rides on the operand load of a hand-written matrix-core kernel
# (csrc/proj_kernels.h::outproj_gate_fwd_kernel, include/hyena_proj.h) instead of cm_post_fwd writing zT for a library GEMM to read back.
# ---------------------------------------------------------------------------------------------------------------------
import os as _os

OUTPROJ_MFMA = _os.environ.get("HYENA_OUTPROJ_MFMA", "1") != "0"      # A/B knob: 0 = cm_post_fwd + library GEMM
# out_proj's input gradient with the gate backward in its epilogue (csrc/proj_kernels.h::outproj_dgrad_gate_bwd_kernel, round 5) against the pair
# it can replace, library GEMM (dz^T) + cm_post_bwd, measured on the MI355X (profiles/r5e_outproj_dgrad.txt): the kernel wins at many short
# sequences (32768 x 8: 190 vs 205 us), ties at 160000 x 2 (224 vs 226) and loses at B = 1 (L = 2^20: 756 - 800 vs 719 us; the pair streams its
# bytes at 4.9 TB/s, the matrix-core kernel at 3.4 - 3.6).  "auto" (default) takes it where it wins (_dgrad_fused); HYENA_OUTPROJ_DGRAD_MFMA=1 / 0 forces it.
DGRAD_MFMA = {"1": True, "0": False}.get(_os.environ.get("HYENA_OUTPROJ_DGRAD_MFMA", "auto"), "auto")


def _dgrad_fused(B, L, D, dtype):
    if DGRAD_MFMA is False or not _lib.outproj_dgrad_supported(B, L, D, dtype):
        return False
    if DGRAD_MFMA is True:
        return True
    # round 6 (profiles/r6o_bench_dgrad.txt, after cm_post_bwd learned to put several short rows into one workgroup): the kernel wins wherever a batch
    # has at least two sequences -- 32768 x 8: 180 vs 193 us, 32767 x 8: 189 vs 227, 160000 x 2: 222 vs 237, 4096 x 64: 180 vs 194 -- except for the
    # many very short rows of the d_model 128 models (1024 x 256 x 128: 105 vs 96 us), and loses at B = 1 (2^20: 938 vs 684)
    if D == 128 and L <= 2048:
        return False
    return B >= 2


def mixer_out_supported(xT, L, out_weight):
    """16-bit channel-major tensors, d_model 128 / 256, sequences of at least one 64-position tile.  (Round 4 switched the kernel off for
    training calls at L % 8 != 0 -- the reference trainer's own lengths -- because zT's rows then started at odd element offsets; with
    pitched rows, _lib.row_pitch, they do not, and the choice no longer depends on L or on the grad mode: ADVICE r4.)"""
    D3, B, Lx = xT.shape
    D = D3 // 3
    return (OUTPROJ_MFMA and xT.dtype in (torch.bfloat16, torch.float16) and tuple(out_weight.shape) == (D, D)
            and (xT.is_cuda or _lib._backend.name != "hip") and _lib.outproj_supported(B, L, Lx, D, xT.dtype))


class HyenaMixerOutCMFunc(torch.autograd.Function):
    """out (B, L, D) = out_proj(fftconv(v * x1, k, bias) * x0): HyenaMixerCMFunc followed by projection.OutProjCMFunc, with the forward's
    ``zT = y * x0`` formed inside the projection kernel (and written out only when out_proj's weight gradient will need it).  The backward
    is the two functions' backward unchanged: dzT and the weight gradient are library GEMMs (contractions over d_model / the positions).

    With ``ln_w`` (round 5): the prenorm block's residual add + LayerNorm behind the mixer (simple_lm.py:280-284; flash_attn's
    dropout_add_layer_norm with p = 0) runs in the same kernel's epilogue -- returns (LayerNorm(residual'), residual') with
    residual' = out_proj(...) + residual in fp32, bit for bit what block.AddLayerNormFunc makes of the unfused output, which is never written."""

    @staticmethod
    def forward(ctx, xT, b_in, sf_weight, sf_bias, k, bias, L, vg, w_out, b_out, residual=None, ln_w=None, ln_b=None, eps=0.0):
        D3, B, Lx = xT.shape
        D = D3 // 3
        xc = _lib.as_cm(xT)
        bi = b_in.detach().to(torch.float32).contiguous()
        w = sf_weight.detach().to(torch.float32).reshape(D3, 3).contiguous()
        b = sf_bias.detach().to(torch.float32).contiguous()
        kf = _lib.as_rows(k.detach().to(torch.float32))
        bf = bias.detach().to(torch.float32).reshape(D).contiguous()
        if vg is None:
            vg = _lib.cm_pre_fwd(xc, bi, w, b, L)
        need = list(_gradmode.needs(ctx)) + [False] * 4
        want_grad = any(need[:6]) or any(need[8:13])
        spectra = None
        if want_grad and _lib.save_spectra_default(B, D, L, device=vg.device):
            y, spectra = _lib.fftconv_fwd(vg, kf, bf, save=True)
        else:
            y = _lib.fftconv_fwd(vg, kf, bf, grad=want_grad)
        wo = _castcache.shadow(w_out, xc.dtype)                                # (an fp32 parameter: its per-step shadow; else a plain cast)
        bo = _castcache.rounded_f32(b_out, xc.dtype)                           # rounded as autocast rounds it, in fp32
        none = torch.empty(0, device=xc.device)
        ctx.norm = ln_w is not None
        if ctx.norm:
            lw = ln_w.detach().to(torch.float32).contiguous()
            lb = ln_b.detach().to(torch.float32).contiguous()
            r2 = None if residual is None else residual.detach().reshape(B * L, D).to(torch.float32).contiguous()
            out, res_out, mean, rstd, zT = _lib.outproj_gate_addnorm_fwd(y, xc, bi, w, b, wo, bo, bool(need[8]), r2, lw, lb, eps)
            ctx.save_for_backward(xc, bi, w, b, kf, bf, y, wo, zT if zT is not None else none, res_out, lw, mean, rstd)
            ctx.norm_meta = (None if residual is None else residual.dtype, ln_w.dtype, ln_b.dtype)
        else:
            out, zT = _lib.outproj_gate_fwd(y, xc, bi, w, b, wo, bo, want_z=bool(need[8]))
            ctx.save_for_backward(xc, bi, w, b, kf, bf, y, wo, zT if zT is not None else none)
        ctx.has_z = zT is not None
        ctx.spectra = spectra
        ctx.meta = (b_in.dtype, sf_weight.shape, sf_weight.dtype, sf_bias.dtype, k.dtype, bias.shape, bias.dtype, L, w_out.dtype,
                    None if b_out is None else b_out.dtype)
        if ctx.norm:
            return out, res_out.view(B, L, D)
        return out

    @staticmethod
    def backward(ctx, dout, dres_out=None):
        from .projection import cm_from_pm, wgrad_pm_cm
        bin_dtype, w_shape, w_dtype, b_dtype, k_dtype, bias_shape, bias_dtype, L, wo_dtype, bo_dtype = ctx.meta
        norm_grads = (None, None, None, None)
        if ctx.norm:
            xc, bi, w, b, kf, bf, y, wo, zT, res_out, lw, mean, rstd = ctx.saved_tensors
            D3, B, Lx = xc.shape
            D = D3 // 3
            rows = B * L
            r_dtype, lw_dtype, lb_dtype = ctx.norm_meta
            # the block's add + LayerNorm backward (block.AddLayerNormFunc.backward): d out_proj output (= dx0) and d residual
            h2 = None if dres_out is None else dres_out.reshape(rows, D).to(torch.float32).contiguous()
            dy2, dres, dlw, dlb = _lib.add_norm_bwd(dout.reshape(rows, D).contiguous(), h2, res_out, lw, mean, rstd, xc.dtype,
                                                    need_dres=r_dtype is not None)
            norm_grads = (None if dres is None else dres.view(B, L, D).to(r_dtype), dlw.to(lw_dtype), dlb.to(lb_dtype), None)
        else:
            xc, bi, w, b, kf, bf, y, wo, zT = ctx.saved_tensors
            D3, B, Lx = xc.shape
            D = D3 // 3
            rows = B * L
            dy2 = dout.to(xc.dtype).reshape(rows, D).contiguous()
        # ---- out_proj's backward (projection.OutProjCMFunc.backward) ----
        dW = dbo = None
        if ctx.needs_input_grad[8]:
            dW = _castcache.wgrad_out(wgrad_pm_cm(dy2, zT if ctx.has_z else _lib.cm_post_fwd(y, xc, bi, w, b)), wo_dtype, xc.dtype)
        if bo_dtype is not None and ctx.needs_input_grad[9]:
            dbo = _castcache.wgrad_out(_lib.colsum(dy2), bo_dtype, xc.dtype)
        if not any(ctx.needs_input_grad[:6]):
            return (None, None, None, None, None, None, None, None, dW, dbo) + norm_grads
        # ---- the core's backward (HyenaMixerCMFunc.backward) ----
        dxT = _lib.empty_like_cm(xc)
        if Lx > L:
            dxT[:, :, L:].zero_()                     # (the kernels write every position < L of every row)
        part = _lib.cm_partials(xc, L)
        part0 = None
        if _dgrad_fused(B, L, D, xc.dtype):
            # dz^T = W_out^T dy^T and the gate's backward in ONE matrix-core kernel: dz^T is never written (csrc/proj_kernels.h, round 5)
            dy, part0 = _lib.outproj_dgrad_gate_bwd(dy2, wo.t().contiguous(), y, xc, bi, w, b, dxT)
        else:
            dzT = cm_from_pm(wo.t(), dy2, B, L)                            # channel-major (pitched rows), straight from the GEMM
            dy = _lib.cm_post_bwd(dzT, y, xc, bi, w, b, dxT, part)
        need_vg = ctx.spectra is None or _lib.lib().hyena_fftconv_plan(int(L)) == _lib.PLAN_ONCHIP
        vg = _lib.cm_pre_fwd(xc, bi, w, b, L) if need_vg else None
        need_dk = ctx.needs_input_grad[4] or ctx.needs_input_grad[5]
        dvg, dk, dbias = _lib.fftconv_bwd(dy, vg, kf, bf, need_du=True, need_dk=need_dk, saved=ctx.spectra)
        ctx.spectra = None
        _lib.cm_pre_bwd(dvg, xc, bi, w, b, dxT, part)
        if part0 is None:
            red = part[:, :, :5].sum(dim=1)
        else:                                                              # channels [0, D): the dgrad kernel's per-run records
            red = torch.cat([part0[:, :, :5].sum(dim=1), part[D:, :, :5].sum(dim=1)], dim=0)
        dw = red[:, :3].reshape(w_shape).to(w_dtype)
        db = red[:, 3].to(b_dtype)
        dbin = red[:, 4].to(bin_dtype)
        return (dxT, dbin, dw, db, dk.to(k_dtype) if dk is not None else None,
                dbias.reshape(bias_shape).to(bias_dtype) if dbias is not None else None, None, None, dW, dbo) + norm_grads


def hyena_mixer_out_cm(xT, b_in, sf_weight, sf_bias, k, bias, L, vg, w_out, b_out, add_norm=None):
    """(3D, B, Lx) -> (B, L, D): the channel-major core and out_proj in one autograd function (see HyenaMixerOutCMFunc); the caller
    checks mixer_out_supported first.  Runs with autocast disabled on tensors already in the compute type, like projection.out_proj_cm.
    add_norm = (residual or None, ln_weight, ln_bias, eps): the prenorm block's residual add + LayerNorm in the same kernel
    -> (LayerNorm(residual'), residual' fp32)."""
    with torch.autocast("cuda" if xT.is_cuda else "cpu", enabled=False):
        if add_norm is not None:
            residual, ln_w, ln_b, eps = add_norm
            return _gradmode.apply(HyenaMixerOutCMFunc, xT, b_in, sf_weight, sf_bias, k, bias, L, vg, w_out, b_out, residual, ln_w, ln_b,
                                   float(eps))
        return _gradmode.apply(HyenaMixerOutCMFunc, xT, b_in, sf_weight, sf_bias, k, bias, L, vg, w_out, b_out)
