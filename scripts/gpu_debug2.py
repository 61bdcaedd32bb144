"""Debug: compare the GPU workspace contents after a forward call with the CPU emulation of the same kernels."""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyena_dna_amd import _lib
dev = torch.device('cuda', 0)
B, D, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
chunk = D
dtype = torch.bfloat16
g = torch.Generator().manual_seed(L + D)
k = torch.randn(D, L, generator=g) * torch.exp(-5.0 * torch.linspace(0, 1, L))[None] * 0.1
bias = torch.randn(D, generator=g)
u = torch.randn(B, D, L, generator=g).to(dtype)
M = _lib.lib().hyena_fftconv_fft_size(L)

def regions(ws):
    f = ws.view(torch.float32)
    n = chunk * M * 2
    return {"Wk": f[:n], "S2": f[n:3 * n], "W": f[3 * n:3 * n + B * n]}

def run_gpu():
    ud, kd, bd = u.to(dev), k.to(dev), bias.to(dev)
    out = _lib.fftconv_fwd(ud, kd, bd, chunk=chunk)
    torch.cuda.synchronize()
    ws = _lib._workspace[(0, torch.cuda.current_stream(dev).cuda_stream)]
    return out.float().cpu(), {n: t.cpu().clone() for n, t in regions(ws).items()}

o1, r1 = run_gpu()
o2, r2 = run_gpu()
print("GPU run1 vs run2: out max diff", (o1 - o2).abs().max().item())
for n in r1:
    d = (r1[n] - r2[n]).abs()
    print(f"  {n}: max diff {d.max().item():.3e}  nonzero {int((d > 0).sum())}")
# CPU emulation of the same kernels
emu = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "hipemu", "_emu_fftconv_test_only.so"))
emu.hyena_fftconv_table_bytes.restype = ctypes.c_size_t
emu.hyena_fftconv_workspace_bytes.restype = ctypes.c_size_t
P = ctypes.c_void_p
tb = torch.zeros(emu.hyena_fftconv_table_bytes(L), dtype=torch.uint8)
assert emu.hyena_fftconv_init_tables(P(tb.data_ptr()), L) == 0
wsb = emu.hyena_fftconv_workspace_bytes(B, D, L, 0, chunk)
ws = torch.zeros(wsb, dtype=torch.uint8)
oe = torch.empty_like(u)
st = emu.hyena_fftconv_fwd(P(u.data_ptr()), P(k.data_ptr()), P(bias.data_ptr()), P(oe.data_ptr()), B, D, L, 1, P(tb.data_ptr()), P(ws.data_ptr()), ctypes.c_size_t(wsb), chunk, None)
assert st == 0
re_ = regions(ws)
print("GPU run1 vs emu: out max diff", (o1 - oe.float()).abs().max().item())
for n in r1:
    d = (r1[n] - re_[n]).abs()
    per = d.view(-1, M * 2 * (2 if n == "S2" else 1)).max(dim=1).values if n != "W" else d.view(B * chunk, -1).max(dim=1).values
    bad = (per > 1e-3 * re_[n].abs().max()).nonzero().flatten().tolist()
    print(f"  {n}: max diff {d.max().item():.3e} (scale {re_[n].abs().max().item():.3e}); bad rows/channels: {bad[:40]} (n={len(bad)})")
    if bad and n != "W":
        c = bad[0]
        dd = d.view(chunk, -1)[c]
        idx = (dd > 1e-3 * re_[n].abs().max()).nonzero().flatten()
        per_el = 4 if n == "S2" else 2
        els = (idx // per_el)
        print(f"     channel {c}: {len(idx)} bad floats; rows (k1) {sorted(set((els // 1024).tolist()))[:40]}; k2 sample {sorted(set((els % 1024).tolist()))[:20]}")

# what ARE the wrong values?  compare with the emulated S2 at neighbouring q (k2 +- 32) of the same row
g2 = r1["S2"].view(chunk, -1, 1024, 4); e2 = re_["S2"].view(chunk, -1, 1024, 4)
dd = (g2 - e2).abs().amax(dim=-1)
bad = (dd > 1e-3 * e2.abs().max()).nonzero()
print("bad S2 elements:", len(bad))
for (c, k1, k2) in bad[:12].tolist():
    cand = {}
    for dq in (-2, -1, 1, 2):
        kk = k2 + 32 * dq
        if 0 <= kk < 1024:
            cand[dq] = (g2[c, k1, k2] - e2[c, k1, kk]).abs().max().item()
    for dr in range(e2.shape[1]):
        if (g2[c, k1, k2] - e2[c, dr, k2]).abs().max().item() < 1e-9 and dr != k1: cand[("row", dr)] = 0.0
    print(f"  ch {c} k1 {k1} k2 {k2} (q {k2 // 32} j {k2 % 32}): gpu {g2[c, k1, k2].tolist()} emu {e2[c, k1, k2].tolist()} | diff to emu@q+dq: { {k: f'{v:.1e}' for k, v in cand.items()} }")
