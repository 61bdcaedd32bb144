"""Round 6 soak: N training steps of a small HyenaDNALM with the filter on a second stream and without -- the parameters after the run must be
bit-identical (same kernels, same operands: a missing wait would show as a race).  python scripts/stream_soak.py L B D n_layer steps"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import hyena_dna_amd.hyena as H  # noqa: E402
import hyena_dna_amd.lm as LM  # noqa: E402

L, B, D, n_layer, steps = (int(x) for x in sys.argv[1:6])
dev = torch.device("cuda", 0)
layer = dict(l_max=L + 2, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10)


def run(side):
    H.FILTER_SIDE_STREAM = side
    torch.manual_seed(0)
    m = LM.HyenaDNALM(d_model=D, n_layer=n_layer, d_inner=4 * D, vocab_size=12, layer=layer, resid_dropout=0.0, embed_dropout=0.0,
                      pad_vocab_size_multiple=8, fused_dropout_add_ln=True, residual_in_fp32=True).to(dev)
    opt = torch.optim.AdamW(m.parameters(), lr=3e-4)
    g = torch.Generator().manual_seed(1)
    losses = []
    for i in range(steps):
        ids = torch.randint(7, 11, (B, L), generator=g).to(dev)
        tgt = torch.roll(ids, -1, 1)
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = m.loss(ids, tgt)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    torch.cuda.synchronize()
    return losses, [p.detach().clone() for p in m.parameters()]


la, pa = run(True)
lb, pb = run(False)
same = la == lb and all(torch.equal(x, y) for x, y in zip(pa, pb))
print(f"L {L} B {B} D {D} layers {n_layer} steps {steps}: losses {la[0]:.4f} -> {la[-1]:.4f}; second stream vs one stream bit-identical: {same}")
assert same
