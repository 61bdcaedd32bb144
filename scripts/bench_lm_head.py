"""The LM head's GEMM at 2^20 tokens: (P, 256) x (256, V) with V = 16 -- how the library does with the output width padded (zeros) to 32 / 64 / 128."""
import torch
dev = torch.device("cuda", 0)
P, K = 1 << 20, 256


def timeit(fn, n=10, w=3):
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


x = torch.randn(P, K, device=dev).to(torch.bfloat16).requires_grad_(True)
for V in (16, 32, 64, 128):
    W = (torch.randn(V, K, device=dev) / 16).to(torch.bfloat16).requires_grad_(True)
    fwd = timeit(lambda: torch.nn.functional.linear(x, W))
    y = torch.nn.functional.linear(x, W)
    dy = torch.randn_like(y)
    bwd = timeit(lambda: torch.autograd.grad(y, (x, W), dy, retain_graph=True))
    print(f"V = {V:3d}: forward {fwd:7.1f} us, backward (dx + dW) {bwd:7.1f} us")
