"""Per-phase timeline of conv_kernel from in-kernel time stamps (profiling build: scripts/build_variant.sh prof -DOC_PROFILE;
run with HYENA_FFTCONV_LIB=build/libhyena_prof.so).  usage: python scripts/oc_phase_profile.py "L B D" ...
Stamps (s_memtime, one set per wavefront): 0 start | 1 pass-1 butterflies done (row loaded, twisted) | 2 pass-1 twiddles applied |
3-6 the four barriers of exchange 1 | 7 pass-2 butterflies | 8 pass-2 twiddles | 9 exchange 2 | 10 pass 3 | 11 filter product |
12 pass 3 (inverse) | 13 exchange 2 | 14 twiddles | 15 butterflies | 16 barrier before exchange 1 | 17-20 its four barriers |
21 twiddles | 22 butterflies | 23 stores issued | 24 stores acknowledged."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyena_dna_amd import _lib  # noqa: E402

NAMES = ["load+twist+dft1", "tw1", "x1 W.re|B", "x1 R.re|B", "x1 W.im|B", "x1 R.im|B", "dft2", "tw2", "x2", "pass3", "H product",
         "pass3^-1", "x2^-1", "tw2^-1", "dft2^-1", "B(pre-x1)", "x1^-1 W.re|B", "x1^-1 R.re|B", "x1^-1 W.im|B", "x1^-1 R.im|B", "tw1^-1",
         "dft1^-1", "untwist+cvt+store issue", "store ack"]
dev = torch.device("cuda", 0)
L_ = _lib.lib()
L_.hyena_oc_prof_set.argtypes = [ctypes.c_void_p]
for cfg in sys.argv[1:]:
    L, B, D = (int(x) for x in cfg.split())
    g = torch.Generator(device=dev).manual_seed(0)
    u = torch.randn(B, D, L, generator=g, device=dev).to(torch.bfloat16)
    k = torch.randn(D, L, generator=g, device=dev) * 0.1
    bias = torch.randn(D, generator=g, device=dev)
    R = max(1, 1 << (max(L, 1024) - 1).bit_length() >> 10)
    waves = max(1, 32 * R // 64)
    rows = B * D
    for _ in range(3):
        _lib.fftconv_fwd(u, k, bias)
    buf = torch.zeros(rows * waves * 32, dtype=torch.int64, device=dev)
    assert L_.hyena_oc_prof_set(ctypes.c_void_p(buf.data_ptr())) == 0
    torch.cuda.synchronize()
    _lib.fftconv_fwd(u, k, bias)
    torch.cuda.synchronize()
    L_.hyena_oc_prof_set(None)
    t = buf.cpu().numpy().reshape(rows, waves, 32).astype(np.int64)
    st = t[:, :, :25]
    clk_per_us = ((st[:, :, 24] - st[:, :, 0]) / np.maximum(1, (t[:, :, 27] - t[:, :, 26])) * 100.0)   # memtime ticks per us (realtime = 100 MHz)
    f = float(np.median(clk_per_us))
    d = np.diff(st, axis=2) / f                                    # us, [row][wave][24]
    tot = (st[:, :, 24] - st[:, :, 0]) / f
    print(f"== L={L} B={B} D={D} R={R}: {rows} workgroups x {waves} wavefronts; s_memtime = {f:.0f} ticks/us; wavefront lifetime "
          f"mean {tot.mean():.2f} us (min {tot.min():.2f}, max {tot.max():.2f})")
    # kernel span and CU occupancy: first start to last end, rows per CU
    span = (st[:, :, 24].max() - st[:, :, 0].min()) / f
    hw = t[:, 0, 25]
    cu_key = ((hw >> 32) << 16) | (hw & 0xff00)            # XCC_ID | SE / SH / CU bits of HW_ID (wave, SIMD, pipe ... masked off)
    ncu = len(np.unique(cu_key))
    print(f"   kernel span {span:.1f} us, {ncu} distinct (xcc, hw_id) values; sum of workgroup lifetimes / span = {tot[:, 0].sum() / span:.1f} workgroups in flight")
    print(f"   {'phase':28s} {'mean':>7s} {'min':>7s} {'max':>7s}   share")
    for i, n in enumerate(NAMES):
        print(f"   {n:28s} {d[:, :, i].mean():7.3f} {d[:, :, i].min():7.3f} {d[:, :, i].max():7.3f}   {100 * d[:, :, i].mean() / tot.mean():5.1f} %")
    # skew between the wavefronts of a workgroup at the first stamps
    sk0 = (st[:, :, 0].max(1) - st[:, :, 0].min(1)) / f
    sk2 = (st[:, :, 2].max(1) - st[:, :, 2].min(1)) / f
    print(f"   wavefront start skew within a workgroup: mean {sk0.mean():.2f} us; arrival skew at exchange 1: {sk2.mean():.2f} us (max {sk2.max():.2f})")
    # gap between consecutive workgroups on the same CU slot: sort by start within equal hw id
    order = np.argsort(st[:, 0, 0])
    ends = {}
    gaps = []
    for r in order:
        key = int(cu_key[r])
        if key in ends:
            gaps.append((st[r, :, 0].min() - ends[key]) / f)
        ends[key] = st[r, :, 24].max()
    if gaps:
        gaps = np.array(gaps)
        print(f"   gap between a workgroup's last stamp and the next workgroup's first on the same (xcc, hw_id): mean {gaps.mean():.2f} us, "
              f"median {np.median(gaps):.2f}, p90 {np.percentile(gaps, 90):.2f}")
