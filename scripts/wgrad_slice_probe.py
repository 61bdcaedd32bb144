"""Which slice length does the library like for the split-K weight-gradient bmm?  (768 x q) @ (q x 256), bf16 -> fp32, batch = slices.
usage: python scripts/wgrad_slice_probe.py"""
import torch

dev = torch.device("cuda", 0)


def timeit(fn, n=20, w=5):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


P = 1048575
g = torch.Generator(device=dev).manual_seed(0)
for C, K, cm in ((768, 256, True), (256, 256, True), (1024, 256, False), (256, 1024, False)):
    # cm: channel-major d (C, P) with pitched rows x position-major x (P, K);  else position-major both: dy (P, C), x (P, K)
    ld = (P + 63) // 64 * 64
    if cm:
        d = torch.randn(C, ld, generator=g, device=dev).bfloat16()[:, :P]
    else:
        d = torch.randn(P, C, generator=g, device=dev).bfloat16()
    x = torch.randn(P, K, generator=g, device=dev).bfloat16()
    for s, q in ((64, 16383), (64, 16128), (64, 16320), (64, 15360), (63, 16384), (127, 8192), (128, 8191), (32, 32768 - 256), (31, 32768), (255, 4096)):
        n = s * q
        if cm:
            a = d[:, :n].reshape(C, s, q).permute(1, 0, 2)
        else:
            a = d[:n].view(s, q, C).transpose(1, 2)
        b = x[:n].view(s, q, K)
        t = timeit(lambda: torch.bmm(a, b, out_dtype=torch.float32).sum(0))
        print(f"C={C} K={K} {'channel-major' if cm else 'position-major'} d: {s} slices of {q}: {t:.1f} us (covers {n} of {P}, rest {P - n})", flush=True)
