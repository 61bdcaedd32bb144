"""Where mlp_kernel's wavefronts spend their time (profiling build: FULL=1 scripts/build_variant.sh pjprof -DPJ_PROFILE; run with
HYENA_FFTCONV_LIB=build/libhyena_pjprof.so).  usage: python scripts/pj_phase_profile.py [P K N]
Phases (s_memtime deltas summed over a wavefront's tiles): 0 wait for my share of the operand tile (vmcnt) | 1 barrier: everybody's share |
2 matrix-core instructions issued | 3 barrier: every wavefront has read its last fragment | 4 next operand tile requested (+ MODE 1: wait for the
a tile) | 5 the epilogue's eight chunks: element-wise work, rows parked in LDS, stores (incl. the wait for the matrix cores; MODE 1: + next a
tile requested) | 9 last stores acknowledged.  (The round-3 kernel and the first LDS-direct version had other phases: profiles/r4j_mlp_phases.txt.)"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyena_dna_amd import _lib  # noqa: E402

NAMES = ["wait: my operand share", "barrier 1 (operand complete)", "MFMA issue", "barrier 2 (operand read)", "request next operand (+ wait a)",
         "epilogue: 8 chunks, row layout (+MFMA drain)", "-", "-", "-", "final store ack"]
dev = torch.device("cuda", 0)
L_ = _lib.lib()
L_.hyena_pj_prof_set.argtypes = [ctypes.c_void_p]
P, K, N = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (1 << 20, 256, 1024)
g = torch.Generator(device=dev).manual_seed(0)
dt = torch.bfloat16
x = torch.randn(P, K, generator=g, device=dev).to(dt)
W1 = (torch.randn(N, K, generator=g, device=dev) / K ** 0.5).to(dt)
b1 = (torch.randn(N, generator=g, device=dev) * 0.1).to(dt).float()
dy = torch.randn(P, K, generator=g, device=dev).to(dt)
W2T = (torch.randn(N, K, generator=g, device=dev) / N ** 0.5).to(dt)


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


a, h = _lib.mlp_fc1_gelu_fwd(x, W1, b1)
for name, fn in (("fc1 + GELU (MODE 0)", lambda: _lib.mlp_fc1_gelu_fwd(x, W1, b1)), ("dh + GELU' (MODE 1)", lambda: _lib.mlp_dh_dgelu_bwd(dy, W2T, a))):
    us = timeit(fn)
    nwg = 8192
    buf = torch.zeros(nwg * 4 * 16, dtype=torch.int64, device=dev)
    assert L_.hyena_pj_prof_set(ctypes.c_void_p(buf.data_ptr())) == 0
    torch.cuda.synchronize()
    fn()
    torch.cuda.synchronize()
    L_.hyena_pj_prof_set(None)
    t = buf.cpu().numpy().reshape(nwg, 4, 16).astype(np.int64)
    live = t[:, 0, 10] > 0
    t = t[live]
    tiles = t[:, :, 10]
    f = float(np.median(t[:, :, 11] / np.maximum(1, t[:, :, 12]) * 100.0))          # s_memtime ticks per us (s_memrealtime = 100 MHz)
    d = t[:, :, :10] / f                                                             # us per wavefront, summed over its tiles
    life = t[:, :, 11] / f
    print(f"== {name}: P={P} K={K} N={N}: {us:.1f} us per launch (instrumented build); {t.shape[0]} workgroups x 4 wavefronts, "
          f"{tiles.mean():.1f} tiles each; s_memtime {f:.0f} ticks/us; wavefront lifetime mean {life.mean():.1f} us (max {life.max():.1f})")
    per_tile = d / tiles[:, :, None]
    print(f"   {'phase':40s} {'us / tile':>10s} {'share':>7s}")
    for i, n in enumerate(NAMES):
        print(f"   {n:40s} {per_tile[:, :, i].mean():10.3f} {100 * d[:, :, i].sum() / life.sum():6.1f} %")
    print(f"   {'sum':40s} {per_tile.sum(2).mean():10.3f}")
    hw = t[:, 0, 13]
    cu = ((hw >> 32) << 16) | (hw & 0xff00)
    wgs_per_cu = np.unique(cu, return_counts=True)[1]
    start = t[:, 0, 14]
    print(f"   {len(wgs_per_cu)} distinct (xcc, se/cu) ids; workgroups per id: min {wgs_per_cu.min()} max {wgs_per_cu.max()}")
