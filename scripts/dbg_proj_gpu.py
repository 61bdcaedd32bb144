"""debug helper: where does vg of the projection kernel differ from cm_pre_fwd on the kernel's own xT?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyena_dna_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
for (B, Lx, Lc, D, dtype) in [(3, 4099, 4000, 128, torch.float16), (3, 4099, 4000, 128, torch.bfloat16), (3, 4099, 4099, 128, torch.float16),
                              (1, 4096, 4096, 128, torch.float16), (2, 70001, 70001, 256, torch.float16)]:
    g = torch.Generator(device=dev).manual_seed(Lx + D)
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)      # noqa: E731
    u = rn(B, Lx, D).to(dtype)
    W = (rn(3 * D, D) / D ** 0.5).to(dtype)
    bin_, w, b = rn(3 * D) * 0.3, rn(3 * D, 3) * 0.5, rn(3 * D) * 0.2
    xT, vg = _lib.inproj_pre_fwd(u, W, bin_, w, b, Lc)
    ref = _lib.cm_pre_fwd(xT, bin_, w, b, Lc)
    bad = (vg != ref).nonzero()
    print((B, Lx, Lc, D, dtype), "mismatches", len(bad), "of", vg.numel())
    if len(bad):
        ls = bad[:, 2]
        print("   l range", ls.min().item(), ls.max().item(), "l % 64 hist", torch.bincount(ls % 64, minlength=64).tolist())
        print("   b hist", torch.bincount(bad[:, 0]).tolist(), "first", bad[:6].tolist())
        for idx in bad[:4].tolist():
            print("   ", idx, vg[tuple(idx)].item(), ref[tuple(idx)].item())
