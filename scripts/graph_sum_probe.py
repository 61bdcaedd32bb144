"""torch column sums replayed from a hipGraph: when do they go wrong?  (development aid; pure PyTorch, none of this repository's kernels)"""
import sys
import torch
dev = torch.device("cuda", 0)
torch.manual_seed(0)
x = torch.randn(8192, 128, device=dev)


def gpu_stress():
    junk = [torch.empty(n, device=dev).normal_() for n in (1000, 100000, 3000000, 17)]
    return all(bool(torch.isfinite(j).all()) for j in junk)


def host_stress():
    return len([bytearray(1 << 14) for _ in range(3000)]) + len([float(k) * 1.5 for k in range(20000)])


def capture(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fn()
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            out = fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    return g, out


def probe(name, fn, stress, n=6):
    g, out = capture(fn)
    prev_ref, log = None, []
    for i in range(n):
        x.normal_()
        if stress:
            stress()
        g.replay()
        torch.cuda.synchronize()
        ref = fn().clone()
        e = float((out - ref).abs().max() / ref.abs().max())
        tag = "ok" if e < 1e-4 else ("STALE" if prev_ref is not None and torch.equal(out, prev_ref) else ("ZERO" if float(out.abs().max()) == 0 else "bad %.2f" % e))
        log.append(tag)
        prev_ref = out.clone()
    print(f"{name:60s} {' '.join(log)}", flush=True)


f32 = lambda: x.sum(0)
bf = lambda: x.to(torch.bfloat16).sum(0, dtype=torch.float32)
mv = lambda: torch.mv(x.t(), torch.ones(x.shape[0], device=dev))
allsum = lambda: x.sum().reshape(1)
probe("x.sum(0) fp32, no eager work in between", f32, None)
probe("x.sum(0) fp32, eager GPU allocations in between", f32, gpu_stress)
probe("x.sum(0) fp32, host allocations in between", f32, host_stress)
probe("bf16 sum(0, dtype=f32), eager GPU allocations in between", bf, gpu_stress)
probe("x.sum() (all), eager GPU allocations in between", allsum, gpu_stress)
probe("mv(x^T, ones), eager GPU allocations in between", mv, gpu_stress)
