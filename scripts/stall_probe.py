"""How often does a training loop stall?  python scripts/stall_probe.py L B D n_layer seconds   (prints the per-step wall-time outliers)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from hyena_dna_amd.lm import HyenaDNALM  # noqa: E402

L, B, D, n_layer, secs = (int(x) for x in sys.argv[1:6])
dev = torch.device("cuda", 0)
layer = dict(l_max=L + 2, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10, lr=6e-4, wd=0.0, lr_pos_emb=0.0)
torch.manual_seed(0)
m = HyenaDNALM(d_model=D, n_layer=n_layer, d_inner=4 * D, vocab_size=12, layer=layer, resid_dropout=0.0, embed_dropout=0.1, pad_vocab_size_multiple=8,
               fused_dropout_add_ln=True, residual_in_fp32=True).to(dev)
opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
ids = torch.randint(7, 11, (B, L), device=dev)
tgt = torch.roll(ids, -1, 1)
times = []
t_end = time.perf_counter() + secs
while time.perf_counter() < t_end:
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = m.loss(ids, tgt)
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    times.append((time.perf_counter() - t0) * 1e3)
ts = sorted(times[3:])
med = ts[len(ts) // 2]
out = [(i, round(t, 1)) for i, t in enumerate(times) if i >= 3 and t > 1.5 * med]
print(f"side={os.environ.get('HYENA_FILTER_SIDE_STREAM', 'auto')} steps {len(times)} median {med:.2f} ms mean(after 3) {sum(times[3:]) / len(times[3:]):.2f} ms; steps slower than 1.5 x median: {out}")
