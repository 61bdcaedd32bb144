import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyena_dna_amd import _lib
from oracle import hyena_oracle as O
dev = torch.device('cuda', 0)
B, D, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]); chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dtype = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(L + D)
rn = lambda *s: torch.randn(*s, generator=g, device=dev)
k = rn(D, L) * torch.exp(-5.0 * torch.linspace(0, 1, L, device=dev))[None] * 0.1
bias = rn(D)
u = rn(B, D, L).to(dtype)
for trial in range(3):
    out = _lib.fftconv_fwd(u, k, bias, chunk=chunk).float()
    ref = O.fftconv_ref(u, k, bias).float()          # torch.fft on the GPU (hipFFT): full-tensor reference
    d = (out - ref).abs()
    bad = d > 0.3
    print(f"trial {trial}: max diff {d.max().item():.4f} max|ref| {ref.abs().max().item():.2f} nbad {int(bad.sum())}")
    if bad.any():
        idx = bad.nonzero()
        print("  b:", sorted(set(idx[:, 0].tolist()))[:20])
        print("  d:", sorted(set(idx[:, 1].tolist()))[:40])
        t = idx[:, 2]
        print("  t range", int(t.min()), int(t.max()), "n2 = (t//2)%1024 sample", sorted(set(((t // 2) % 1024).tolist()))[:20], "n1 sample", sorted(set((t // 2048).tolist()))[:20])
        i = idx[0]; print("  first", i.tolist(), out[tuple(i)].item(), ref[tuple(i)].item())
