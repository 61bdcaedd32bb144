"""Which piece of the training step misbehaves when replayed from a hipGraph with eager work in between?  (development aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from hyena_dna_amd.lm import HyenaDNALM, Mlp, GPT2Embeddings
from hyena_dna_amd.hyena import HyenaOperator
from hyena_dna_amd.block import dropout_add_layer_norm
dev = torch.device("cuda", 0)
L, B, D = 1024, 8, 128
torch.manual_seed(0)


def stress(i):
    junk = [torch.empty(n, device=dev).normal_() for n in (1000, 100000, 3000000, 17)]
    ok = all(bool(torch.isfinite(j).all()) for j in junk)
    host = [bytearray(1 << 14) for _ in range(3000)]
    host2 = [float(k) * 1.5 for k in range(20000)]
    return ok and len(host) + len(host2) > 0


def check(name, fn, n=6):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fn()
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            outs = fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    ref = [o.detach().clone() for o in fn()]
    torch.cuda.synchronize()
    worst, bad = 0.0, []
    for i in range(n):
        stress(i)
        g.replay()
        torch.cuda.synchronize()
        for j, (o, r) in enumerate(zip(outs, ref)):
            o = o.detach().float(); r = r.float()
            if not torch.isfinite(o).all():
                bad.append((i, j, "nonfinite"))
                continue
            e = float((o - r).abs().max() / (r.abs().max() + 1e-30))
            worst = max(worst, e)
            if e > 1e-2:
                bad.append((i, j, "%.2e" % e))
    print(f"{name:28s} worst rel err {worst:.2e}  bad {bad[:6]}", flush=True)


x = torch.randn(B, L, D, device=dev)
dy = torch.randn(B, L, D, device=dev)

mlp = Mlp(D, 4 * D, activation=lambda t: F.gelu(t, approximate="tanh")).to(dev)
def f_mlp():
    xx = x.clone().requires_grad_(True)
    for p in mlp.parameters(): p.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = mlp(xx)
    y.backward(dy.to(y.dtype))
    return [y, xx.grad] + [p.grad for p in mlp.parameters()]
check("Mlp fwd+bwd", f_mlp)

lin = torch.nn.Linear(D, 4 * D).to(dev)
def f_lin():
    xx = x.clone().requires_grad_(True)
    for p in lin.parameters(): p.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = lin(xx)
    y.backward(torch.ones_like(y))
    return [y, xx.grad] + [p.grad for p in lin.parameters()]
check("nn.Linear fwd+bwd (torch)", f_lin)

def f_sum():
    t = dy.to(torch.bfloat16).reshape(-1, D)
    return [t.sum(0, dtype=torch.float32), t.float().sum(0)]
check("column sums", f_sum)

op = HyenaOperator(d_model=D, l_max=L + 2, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10).to(dev)
def f_op():
    xx = x.clone().requires_grad_(True)
    for p in op.parameters(): p.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = op(xx)
    y.backward(dy.to(y.dtype))
    return [y, xx.grad] + [p.grad for p in op.parameters() if p.grad is not None]
check("HyenaOperator fwd+bwd", f_op)

w = torch.randn(D, device=dev, requires_grad=True); b = torch.randn(D, device=dev, requires_grad=True)
def f_ln():
    xx = x.to(torch.bfloat16).clone().requires_grad_(True); rr = dy.clone().requires_grad_(True)
    w.grad = None; b.grad = None
    o, r = dropout_add_layer_norm(xx, rr, w, b, 0.0, 1e-5, prenorm=True, residual_in_fp32=True)
    (o.float().sum() + (r * r).sum()).backward()
    return [o, r, xx.grad, rr.grad, w.grad, b.grad]
check("add + LayerNorm fwd+bwd", f_ln)

layer = dict(l_max=L + 2, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10)
model = HyenaDNALM(d_model=D, n_layer=2, d_inner=4 * D, vocab_size=12, layer=layer, resid_dropout=0.0, embed_dropout=0.0,
                   pad_vocab_size_multiple=8).to(dev)
ids = torch.randint(7, 11, (B, L), device=dev); tgt = torch.roll(ids, -1, 1)
def f_model():
    for p in model.parameters(): p.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = model.loss(ids, tgt)
    loss.backward()
    return [loss] + [p.grad for p in model.parameters()]
check("2-layer model fwd+bwd", f_model)

params = [torch.randn(300, 300, device=dev, requires_grad=True) for _ in range(40)]
for p in params: p.grad = torch.randn_like(p)
opt = torch.optim.AdamW(params, lr=1e-3, capturable=True)
state0 = [p.detach().clone() for p in params]
def f_opt():
    with torch.no_grad():
        for p, s in zip(params, state0): p.copy_(s)
    opt.step()
    return [p for p in params[:4]]
# the optimizer's own state advances at every call, so only finiteness is meaningful here
check("AdamW capturable step", f_opt)
