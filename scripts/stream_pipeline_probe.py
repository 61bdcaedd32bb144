#!/usr/bin/env python
"""Diagnostic: does running several small-chunk kernel chains concurrently (one per stream, each on its own channel
slice) keep the intermediates in the Infinity Cache and beat one whole-D chain?  L = 2^20, d = 256, B = 1, forward."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hyena_dna_amd import _lib

dev = torch.device("cuda", 0)
L, D = 1 << 20, 256
u = torch.randn(1, D, L, device=dev).bfloat16()
k = torch.randn(D, L, device=dev) * 0.01
bias = torch.randn(D, device=dev)


def run(nstreams, chunk, reps=5):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    per = D // nstreams

    def once():
        cur = torch.cuda.current_stream()
        for i, s in enumerate(streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                _lib.fftconv_fwd(u[:, i * per:(i + 1) * per], k[i * per:(i + 1) * per], bias[i * per:(i + 1) * per], chunk=chunk)
        for s in streams:
            cur.wait_stream(s)

    once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        once()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


dout = torch.randn(1, D, L, device=dev).bfloat16()


def run_fb(nstreams, reps=5):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    per = D // nstreams

    def once():
        cur = torch.cuda.current_stream()
        for i, s in enumerate(streams):
            s.wait_stream(cur)
            sl = slice(i * per, (i + 1) * per)
            with torch.cuda.stream(s):
                out, saved = _lib.fftconv_fwd(u[:, sl], k[sl], bias[sl], save=True)
                _lib.fftconv_bwd(dout[:, sl], u[:, sl], k[sl], bias[sl], saved=saved)
        for s in streams:
            cur.wait_stream(s)

    once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        once()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for ns in (1, 2, 4, 8, 16):
    print(f"fwd+bwd, {ns} streams (whole slices): {run_fb(ns):.3f} ms")
base = run(1, 256)
print(f"1 stream, whole D: {base:.3f} ms (forward)")
for ns in (2, 4):
    for ch in (16, 32, 64, 128):
        if ch > D // ns:
            continue
        print(f"{ns} streams x chunk {ch:3d}: {run(ns, ch):.3f} ms")
