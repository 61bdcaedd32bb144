"""TEST HELPER (not product code): the smallest stack that wires every piece of this package together the way the
reference's backbone does (src/models/sequence/long_conv_lm.py:340-396, simple_lm.py:259-290): token embedding ->
N x [add+LayerNorm -> HyenaOperator -> add+LayerNorm -> MLP] -> add+LayerNorm -> tied LM head.  Used by the training
smoke tests to check that gradients flow through all the autograd Functions together."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from hyena_dna_amd.block import dropout_add_layer_norm
from hyena_dna_amd.hyena import HyenaOperator
from hyena_dna_amd.projection import hyena_linear


class TinyBlock(nn.Module):
    def __init__(self, d, l_max):
        super().__init__()
        self.norm1, self.norm2 = nn.LayerNorm(d), nn.LayerNorm(d)
        self.mixer = HyenaOperator(d_model=d, l_max=l_max, order=2, filter_order=64, emb_dim=5, short_filter_order=3,
                                   modulate=True, w=10, lr=6e-4, wd=0.0, lr_pos_emb=0.0)
        self.fc1, self.fc2 = nn.Linear(d, 4 * d), nn.Linear(4 * d, d)

    def forward(self, h, residual):
        h, residual = dropout_add_layer_norm(h, residual, self.norm1.weight, self.norm1.bias, 0.0, self.norm1.eps, prenorm=True,
                                             residual_in_fp32=True)
        h = self.mixer(h)
        h, residual = dropout_add_layer_norm(h, residual, self.norm2.weight, self.norm2.bias, 0.0, self.norm2.eps, prenorm=True,
                                             residual_in_fp32=True)
        h = hyena_linear(F.gelu(hyena_linear(h, self.fc1.weight, self.fc1.bias), approximate="tanh"), self.fc2.weight, self.fc2.bias)
        return h, residual


class TinyLM(nn.Module):
    def __init__(self, vocab, d, l_max, n_layer):
        super().__init__()
        self.emb = nn.Embedding(vocab, d)
        self.blocks = nn.ModuleList([TinyBlock(d, l_max) for _ in range(n_layer)])
        self.ln_f = nn.LayerNorm(d)

    def forward(self, ids):
        h, residual = self.emb(ids), None
        for blk in self.blocks:
            h, residual = blk(h, residual)
        h = dropout_add_layer_norm(h, residual, self.ln_f.weight, self.ln_f.bias, 0.0, self.ln_f.eps, prenorm=False,
                                   residual_in_fp32=True)
        return F.linear(h, self.emb.weight)                 # tied head (long_conv_lm.py:462-465)


def train(device, steps, d=64, L=256, B=4, n_layer=2, autocast_dtype=None, seed=0, lr=3e-3, warmup=0):
    """next-nucleotide prediction on periodic synthetic DNA (period 7) through the vectorised tokenizer; returns the losses.
    `warmup`: the learning rate ramps linearly to `lr` over that many updates (0: constant)"""
    from hyena_dna_amd.tokenizer import DNACharTokenizerLUT
    torch.manual_seed(seed)
    tok = DNACharTokenizerLUT()
    motif = "ACGTTGA"
    seqs = [(motif * (L // len(motif) + 2))[s:s + L] for s in range(B)]
    data, target = zip(*(tok.sample(s, L + 1, add_eos=True) for s in seqs))
    data, target = torch.stack(data).to(device), torch.stack(target).to(device)
    model = TinyLM(16, d, L, n_layer).to(device)
    opt = torch.optim.AdamW(model.parameters(), lr=lr)
    losses = []
    for it in range(steps):
        if warmup:
            for g in opt.param_groups:
                g["lr"] = lr * min(1.0, (it + 1) / warmup)
        opt.zero_grad(set_to_none=True)
        with torch.autocast(device_type="cuda" if device != "cpu" else "cpu", dtype=autocast_dtype or torch.bfloat16,
                            enabled=autocast_dtype is not None):
            logits = model(data)
        loss = F.cross_entropy(logits.float().reshape(-1, 16), target.reshape(-1))
        loss.backward()
        opt.step()
        losses.append(loss.item())
    return losses
