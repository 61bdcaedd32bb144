#!/usr/bin/env python
"""The reference's own long-convolution path executed on the MI355X: unfused torch.fft (hipFFT) + elementwise kernels.

This is the "same algorithm, same hardware, no fusion" comparison SURVEY.md 8d recommends next to the CPU baseline.  It
restates `fftconv_ref` (reference src/models/sequence/hyena.py:59-88: N = 2L, rfft(k)/N, rfft(u.float()), product,
irfft(norm="forward")[..., :L], + u * bias, cast back) in plain torch ops and times forward + backward through
autograd with HIP events.  It is a measurement aid only: nothing in the package imports it.

    python scripts/bench_unfused_gpu.py L B D [dtype]   ->  one JSON line
"""
import json
import sys

import torch


def unfused(u, k, bias):
    L = u.shape[-1]
    n = 2 * L
    k_f = torch.fft.rfft(k, n=n) / n
    u_f = torch.fft.rfft(u.to(k.dtype), n=n)
    y = torch.fft.irfft(u_f * k_f, n=n, norm="forward")[..., :L]
    return (y + u * bias.unsqueeze(-1)).to(u.dtype)


def main():
    L, B, D = (int(a) for a in sys.argv[1:4])
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[sys.argv[4] if len(sys.argv) > 4 else "bf16"]
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    u = torch.randn(B, D, L, generator=g).to(dev, dtype).requires_grad_()
    t = torch.linspace(0, 1, L)
    k = (torch.randn(D, L, generator=g) * torch.exp(-5 * t) * 0.1).to(dev).requires_grad_()
    bias = torch.randn(D, generator=g).to(dev).requires_grad_()
    dout = torch.randn(B, D, L, generator=g).to(dev, dtype)

    def step():
        u.grad = k.grad = bias.grad = None
        unfused(u, k, bias).backward(dout)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    steps = 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    print(json.dumps({"what": "unfused torch.fft (hipFFT) fftconv_ref fwd+bwd on GPU", "seq_len": L, "batch": B, "d_model": D,
                      "dtype": str(dtype).split(".")[-1], "ms_per_step": ms, "nt_per_s": B * L / ms * 1e3,
                      "peak_mem_GB": torch.cuda.max_memory_allocated() / 2**30}))


if __name__ == "__main__":
    main()
