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
import torch
import torch.nn.functional as F

__all__ = ["hyena_linear", "SplitKLinearFunc", "split_count", "split_k_weight_grad"]

MIN_ROWS = 32768          # below this the library's own schedule is fine
MAX_SPLITS = 64
MIN_SLICE = 4096


def split_count(rows):
    """Number of equal position slices of the batched weight-gradient GEMM: the largest power of two <= MAX_SPLITS with
    slices of >= MIN_SLICE rows.  ``rows`` need not be divisible by it: the ``rows % S`` leftover rows go through one small
    tail GEMM (the reference dataset yields L = max_length - 1 -- 999,999 / 449,999 / 159,999 / 32,767,
    hg38_dataset.py:220 -- so B*L is odd in real training)."""
    s = 1
    while s < MAX_SPLITS and rows // (2 * s) >= MIN_SLICE:
        s *= 2
    return s


def split_k_weight_grad(dy2, x2):
    """dy2^T x2 (fp32) as S batched slices + a tail, partial sums added in a fixed order (deterministic)."""
    rows, n = dy2.shape
    k = x2.shape[1]
    s = split_count(rows)
    body = (rows // s) * s
    a, b = dy2[:body].view(s, rows // s, n).transpose(1, 2), x2[:body].view(s, rows // s, k)
    if dy2.is_cuda:
        part = torch.bmm(a, b, out_dtype=torch.float32)             # fp32 partial sums straight out of the MFMA accumulators
    else:
        part = torch.bmm(a.float(), b.float())                      # host tensors (unit tests): bmm has no out_dtype there
    dw = part.sum(0)
    if body < rows:                                                 # < S leftover rows
        dw = dw + torch.mm(dy2[body:].t().float(), x2[body:].float())
    return dw


class SplitKLinearFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        n, k = weight.shape
        dy2 = dy.reshape(-1, n)
        x2 = x.reshape(-1, k)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.mm(dy2, weight).view(x.shape)
        if ctx.needs_input_grad[1]:
            dw = split_k_weight_grad(dy2, x2).to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy2.sum(0, dtype=torch.float32).to(dy.dtype)
        return dx, dw, db


def hyena_linear(x, weight, bias):
    """``F.linear(x, weight, bias)`` with the autocast semantics of ``nn.Linear`` and the split-K weight gradient."""
    if x.is_cuda and x.shape[-1] == weight.shape[1] and x.numel() // x.shape[-1] >= MIN_ROWS:
        if torch.is_autocast_enabled():
            dt = torch.get_autocast_dtype("cuda")
            if dt in (torch.bfloat16, torch.float16):
                with torch.autocast("cuda", enabled=False):
                    return SplitKLinearFunc.apply(x.to(dt).contiguous(), weight.to(dt), None if bias is None else bias.to(dt))
        elif x.dtype == weight.dtype and x.dtype in (torch.float32, torch.bfloat16, torch.float16):
            return SplitKLinearFunc.apply(x.contiguous(), weight, bias)          # plain fp32 / 16-bit training: same pathology
    return F.linear(x, weight, bias)
