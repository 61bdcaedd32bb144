import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hyena_dna_amd.lm import HyenaDNALM, GraphedTrainStep
dev = torch.device("cuda", 0)
L, B, D, NL = 1024, 8, 128, 8
torch.manual_seed(0)
layer = dict(l_max=L + 2, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10, lr=6e-4, wd=0.0, lr_pos_emb=0.0)
model = HyenaDNALM(d_model=D, n_layer=NL, d_inner=4 * D, vocab_size=12, layer=layer, resid_dropout=0.0, embed_dropout=0.1,
                   pad_vocab_size_multiple=8, fused_dropout_add_ln=True, residual_in_fp32=True).to(dev)
g = torch.Generator(device=dev).manual_seed(2222)
ids = torch.randint(7, 11, (B, L), generator=g, device=dev); tgt = torch.roll(ids, -1, 1)
neager = int(sys.argv[1]) if len(sys.argv) > 1 else 7
cap_first = len(sys.argv) > 2 and sys.argv[2] == "cap"
opt = torch.optim.AdamW(model.parameters(), lr=6e-4, weight_decay=0.1, capturable=cap_first)
for i in range(neager):
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = model.loss(ids, tgt)
    loss.backward(); opt.step()
if neager:
    print("eager loss", float(loss.detach()))
    del loss
diag = os.environ.get("DIAG", "1") == "1"
opt_g = opt if cap_first else torch.optim.AdamW(model.parameters(), lr=6e-4, weight_decay=0.1, capturable=True)
st = GraphedTrainStep(model, opt_g, ids, tgt, warmup=2)
nrep = int(os.environ.get("NREP", "4"))
if not diag:
    for i in range(nrep):
        l = st()
    torch.cuda.synchronize()
    print("final loss after", nrep, "replays:", float(l))
    bad_p = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
    print("nonfinite params", len(bad_p), bad_p[:4])
    sys.exit(0)
kind = os.environ.get("STRESS", "grads")
for i in range(nrep):
    l = float(st())
    if kind == "alloc":
        junk = [torch.empty(n, device=dev).normal_() for n in (1000, 100000, 3000000, 17)]
        ok = all(bool(torch.isfinite(j).all()) for j in junk)
        print(f"replay {i}: loss {l:.4f}", flush=True)
        continue
    bad_g = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    bad_p = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
    none_g = [n for n, p in model.named_parameters() if p.grad is None]
    print(f"replay {i}: loss {l:.4f} nonfinite grads {len(bad_g)} {bad_g[:3]} params {len(bad_p)} {bad_p[:3]} none-grads {len(none_g)}", flush=True)
