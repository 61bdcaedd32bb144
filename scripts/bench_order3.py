"""Round 6: one HyenaOperator layer of order 3 (configs/model/layer/hyena_dna.yaml), fwd + bwd under bf16 autocast -- the channel-major route
(mixer.HyenaMixerCMOrderNFunc) against the generic route (the reference's graph around the HIP convolution): python scripts/bench_order3.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import hyena_dna_amd.hyena as H  # noqa: E402

dev = torch.device("cuda", 0)


def time_route(op, u, dy, steps=6):
    def step():
        op.zero_grad(set_to_none=True)
        ud = u.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = op(ud)
        y.backward(dy)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        step()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts), sorted(ts)[len(ts) // 2], torch.cuda.max_memory_allocated() / 2 ** 30


for B, L, D, order in [(1, 1 << 20, 256, 3), (1, 1048575, 256, 3), (8, 32767, 256, 3), (256, 1023, 128, 3), (1, 1 << 20, 256, 2)]:
    torch.manual_seed(0)
    op = H.HyenaOperator(d_model=D, l_max=L + 2, order=order, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10).to(dev)
    u = torch.randn(B, L, D, device=dev).to(torch.bfloat16)
    dy = torch.randn(B, L, D, device=dev).to(torch.bfloat16)
    line = f"B {B} L {L} D {D} order {order}:"
    for fused in ((True, False) if order > 2 else (True,)):
        H.ORDER_N_FUSED = fused
        torch.cuda.reset_peak_memory_stats()
        try:
            mn, med, mem = time_route(op, u, dy)
            line += f"  {op._route(L)} min {mn:.3f} ms median {med:.3f} ms peak {mem:.1f} GiB;"
        except torch.OutOfMemoryError:
            line += f"  {op._route(L)} out of memory;"
        torch.cuda.empty_cache()
    print(line, flush=True)
    del op, u, dy
    torch.cuda.empty_cache()
