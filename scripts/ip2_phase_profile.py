"""Where in_proj generation 2's wavefronts spend their time (profiling build: scripts/build_proj_variant.sh ip2prof -DPJ_PROFILE [...]; run with
HYENA_FFTCONV_LIB=build/libhyena_ip2prof.so).  Phases per loop iteration (s_memtime deltas summed over a wavefront's tiles):
0 wait for my share of the operand tile | 1 barrier 1 | 2 next operand tile requested | 3 product + the previous tile's row phase | 4 barrier 2 | 5 halo + park"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyena_dna_amd import _lib  # noqa: E402

NAMES = ["wait: my operand share (vmcnt)", "barrier 1", "request next operand tile", "product + previous row phase", "barrier 2", "halo + park"]
dev = torch.device("cuda", 0)
L_ = _lib.lib()
L_.hyena_pj_prof_set.argtypes = [ctypes.c_void_p]
L, B, D = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (1 << 20, 1, 256)
g = torch.Generator(device=dev).manual_seed(0)
u = torch.randn(B, L, D, generator=g, device=dev).to(torch.bfloat16)
W = (torch.randn(3 * D, D, generator=g, device=dev) / D ** 0.5).to(torch.bfloat16)
bin_, w, b = torch.randn(3 * D, generator=g, device=dev), torch.randn(3 * D, 3, generator=g, device=dev), torch.randn(3 * D, generator=g, device=dev)
_lib.proj_kernel_generation(1, 2)
for _ in range(3):
    _lib.inproj_pre_fwd(u, W, bin_, w, b, L)
torch.cuda.synchronize()
nwg, NW = 4096, 12
buf = torch.zeros(nwg * NW * 16, dtype=torch.int64, device=dev)
assert L_.hyena_pj_prof_set(ctypes.c_void_p(buf.data_ptr())) == 0
torch.cuda.synchronize()
_lib.inproj_pre_fwd(u, W, bin_, w, b, L)
torch.cuda.synchronize()
L_.hyena_pj_prof_set(None)
t = buf.cpu().numpy().reshape(nwg, NW, 16).astype(np.int64)
t = t[t[:, 0, 10] > 0]
tiles = t[:, :, 10]
f = float(np.median(t[:, :, 11] / np.maximum(1, t[:, :, 12]) * 100.0))
d = t[:, :, :10] / f
life = t[:, :, 11] / f
print(f"in_proj generation 2, L={L} B={B} D={D}: {t.shape[0]} workgroups x {NW} wavefronts, {tiles.mean():.1f} iterations each; s_memtime {f:.0f} ticks/us; "
      f"wavefront lifetime mean {life.mean():.1f} us (max {life.max():.1f})")
for grp, nm in ((slice(0, 4), "x0 wavefronts"), (slice(4, 8), "x1 wavefronts"), (slice(8, 12), "v wavefronts")):
    pt = d[:, grp, :] / tiles[:, grp, None]
    print(f" {nm}: " + " | ".join(f"{n}: {pt[:, :, i].mean():.3f}" for i, n in enumerate(NAMES)) + f" | sum {pt[:, :, :6].sum(2).mean():.3f} us / iteration")
