"""What the part sustains for write-dominated streams of the MLP kernels' size (torch fill / copy kernels, 16-byte accesses)."""
import torch
dev = torch.device("cuda", 0)


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


for gb in (0.54, 2.15, 4.3):
    n = int(gb * 1e9 / 2)
    x = torch.empty(n, dtype=torch.bfloat16, device=dev)
    y = torch.empty(n, dtype=torch.bfloat16, device=dev)
    us = timeit(lambda: x.zero_())
    print(f"zero_ {gb:.2f} GB: {us:.1f} us = {gb * 1e3 / us:.2f} TB/s written")
    us = timeit(lambda: x.fill_(1.5))
    print(f"fill_ {gb:.2f} GB: {us:.1f} us = {gb * 1e3 / us:.2f} TB/s written")
    us = timeit(lambda: y.copy_(x))
    print(f"copy_ {gb:.2f} GB -> {gb:.2f} GB: {us:.1f} us = {2 * gb * 1e3 / us:.2f} TB/s read + written")
    us = timeit(lambda: torch.mul(x, 2.0, out=y))
    print(f"mul   {gb:.2f} GB -> {gb:.2f} GB: {us:.1f} us = {2 * gb * 1e3 / us:.2f} TB/s read + written")
# one read, four writes (the forward MLP kernel's mix: x -> a, h of 4x the width)
n = int(0.54e9 / 2)
x = torch.empty(n, dtype=torch.bfloat16, device=dev)
outs = [torch.empty(n, dtype=torch.bfloat16, device=dev) for _ in range(8)]
def one_to_eight():
    for o in outs:
        torch.mul(x, 2.0, out=o)
us = timeit(one_to_eight)
print(f"8 x (0.54 GB -> 0.54 GB) back to back: {us:.1f} us = {16 * 0.54 * 1e3 / us:.2f} TB/s")
