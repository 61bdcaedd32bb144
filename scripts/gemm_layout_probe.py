"""hipBLASLt time of the operator's projections in the two activation layouts (B L, C) "position-major" vs (C, B L) "channel-major",
forward and both gradients, bf16, K = D = 256.  usage: python scripts/gemm_layout_probe.py [BL]"""
import sys
import torch

BL = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
D = 256
dev = torch.device("cuda", 0)
dt = torch.bfloat16


def t(fn, n=10, w=3):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for N in (768, 256):
    W = torch.randn(N, D, device=dev, dtype=dt) * 0.05
    b = torch.randn(N, device=dev, dtype=dt)
    u = torch.randn(BL, D, device=dev, dtype=dt)              # position-major input
    uT = torch.randn(D, BL, device=dev, dtype=dt)             # channel-major input
    dy = torch.randn(BL, N, device=dev, dtype=dt)
    dyT = torch.randn(N, BL, device=dev, dtype=dt)
    S = 64
    print(f"N={N} BL={BL}")
    print("  fwd  pos->pos  x = u W^T + b           %.3f ms" % t(lambda: torch.addmm(b, u, W.t())))
    print("  fwd  pos->chan xT = W u^T + b[:,None]  %.3f ms" % t(lambda: torch.addmm(b[:, None], W, u.t())))
    print("  fwd  chan->pos y = uT^T W^T + b        %.3f ms" % t(lambda: torch.addmm(b, uT.t(), W.t())))
    print("  dx   pos:  du = dy W                   %.3f ms" % t(lambda: torch.mm(dy, W)))
    print("  dx   chan->pos: du = dyT^T W           %.3f ms" % t(lambda: torch.mm(dyT.t(), W)))
    print("  dx   pos->chan: duT = W^T dy^T         %.3f ms" % t(lambda: torch.mm(W.t(), dy.t())))
    print("  dW   pos (split-K bmm)                 %.3f ms" % t(lambda: torch.bmm(dy.view(S, BL // S, N).transpose(1, 2), u.view(S, BL // S, D), out_dtype=torch.float32).sum(0)))
    print("  dW   chan dyT, pos u (split-K bmm)     %.3f ms" % t(lambda: torch.bmm(dyT.view(N, S, BL // S).permute(1, 0, 2), u.view(S, BL // S, D), out_dtype=torch.float32).sum(0)))
    print("  dW   pos dy, chan uT (split-K bmm)     %.3f ms" % t(lambda: torch.bmm(dy.view(S, BL // S, N).transpose(1, 2), uT.view(D, S, BL // S).permute(1, 2, 0), out_dtype=torch.float32).sum(0)))
    print("  db   pos  dy.sum(0)                    %.3f ms" % t(lambda: dy.sum(0, dtype=torch.float32)))
    print("  db   chan dyT.sum(1)                   %.3f ms" % t(lambda: dyT.sum(1, dtype=torch.float32)))
