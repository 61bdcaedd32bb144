"""Round 6 diagnostic: where dW1 = da^T x (1024 x 256 outputs) loses 100 us at 2^20 - 1 rows against 2^20: first-level slices of 16320 / 16128 / 16384 rows,
the 256-row second level, the masked tail -- each piece timed alone.  python scripts/wgrad_piece_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from hyena_dna_amd import projection as P  # noqa: E402

dev = torch.device("cuda", 0)


def timeit(fn, n=30, w=5):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


rows = 1048575
for N, K in ((1024, 256), (256, 1024), (256, 256)):
    da = torch.randn(rows, N, device=dev).to(torch.bfloat16)
    x = torch.randn(rows, K, device=dev).to(torch.bfloat16)
    out = [f"N {N} K {K}:"]
    for s, q in ((64, 16384 - 64), (64, 16320), (64, 16128), (64, 15872), (63, 16640), (32, 32640), (128, 8128)):
        f = lambda: P._bmm_f32(da[:s * q].view(s, q, N).transpose(1, 2), x[:s * q].view(s, q, K)).sum(0)      # noqa: E731
        out.append(f"{s} x {q}: {timeit(f):.1f}")
    p0 = 64 * 16320
    f2 = lambda: P._bmm_f32(da[p0:p0 + 15 * 256].view(15, 256, N).transpose(1, 2), x[p0:p0 + 15 * 256].view(15, 256, K)).sum(0)      # noqa: E731
    out.append(f"second level 15 x 256: {timeit(f2):.1f}")
    f3 = lambda: P._tail_product(da.t(), x, rows - 255)      # noqa: E731
    out.append(f"masked tail: {timeit(f3):.1f}")
    f4 = lambda: P.split_k_weight_grad(da, x)      # noqa: E731
    out.append(f"whole: {timeit(f4):.1f}")
    da2, x2 = da[:1 << 20].contiguous() if False else torch.randn(1 << 20, N, device=dev).to(torch.bfloat16), torch.randn(1 << 20, K, device=dev).to(torch.bfloat16)
    out.append(f"whole at 2^20: {timeit(lambda: P.split_k_weight_grad(da2, x2)):.1f}")
    print("  ".join(out), flush=True)
    del da, x, da2, x2
    torch.cuda.empty_cache()
