#!/usr/bin/env python
"""Mint the language-model golden from the REAL reference (imported from /root/reference): ``SimpleLMHeadModel``
(src/models/sequence/simple_lm.py:26-305 -- the reference's own PyTorch restatement of the flash_attn backbone) with the
reference ``HyenaOperator`` as mixer, a hyenadna-tiny-shaped stack (d_model 128, 2 layers, d_inner 512, vocab 12 padded to
16) at L = 4096 (the workspace-free long-convolution plan, whole 16-byte vectors in the shell kernels), fp32 on the CPU:
logits, loss and EVERY parameter gradient.

    python oracle/make_golden_lm.py            # rewrites tests/golden/lm_simple_d128_l4096.pt   (build container only)

TEST INFRASTRUCTURE: the fixture pins ``hyena_dna_amd.lm.HyenaDNALM`` on the GPU (tests/test_gpu_contract.py); nothing in the
product imports this file.  Stubs as in oracle/make_golden.py (they touch no arithmetic).
"""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import OUT, import_reference  # noqa: E402

NAME = "lm_simple_d128_l4096.pt"
CFG = dict(d_model=128, n_layer=2, d_inner=512, vocab_size=12, resid_dropout=0.0, embed_dropout=0.0, pad_vocab_size_multiple=8,
           residual_in_fp32=True)
L, B = 4096, 2
LAYER = dict(_name_="hyena", l_max=L + 2, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10, lr=6e-4,
             wd=0.0, lr_pos_emb=0.0)


def import_simple_lm():
    import_reference()
    import transformers.tokenization_utils  # noqa: F401  (probes torchvision; must come before the stub below)

    class _SD(torch.nn.Module):                       # torchvision.ops.StochasticDepth with p = 0: identity
        def __init__(self, p, mode):
            super().__init__()

        def forward(self, x):
            return x
    ops = types.ModuleType("torchvision.ops")
    ops.StochasticDepth = _SD
    tv = types.ModuleType("torchvision")
    tv.ops = ops
    sys.modules["torchvision"], sys.modules["torchvision.ops"] = tv, ops
    import src.models.sequence.simple_lm as ref_simple
    return ref_simple


def main():
    ref_simple = import_simple_lm()
    torch.manual_seed(20240924)
    model = ref_simple.SimpleLMHeadModel(layer=dict(LAYER), **CFG)
    # the reference initialises every bias to zero (long_conv_lm.py:204-246); give them values so that their gradients and
    # their place in the forward are pinned too
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith(".bias") and "filter_fn.bias" not in n and "norm" not in n and "ln_f" not in n:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    ids = torch.randint(7, 11, (B, L), generator=g)           # A, C, G, T (hg38_char_tokenizer.py:59-66)
    tgt = torch.roll(ids, -1, 1)
    logits = model(ids)[0].logits
    loss = torch.nn.functional.cross_entropy(logits.float().reshape(-1, logits.shape[-1]), tgt.reshape(-1))
    loss.backward()
    out = dict(cfg=CFG, layer={k: v for k, v in LAYER.items() if k != "_name_"}, L=L, B=B,
               state_dict={k: v.detach().clone() for k, v in model.state_dict().items()},
               ids=ids, targets=tgt, logits=logits.detach().clone(), loss=float(loss),
               grads={n: p.grad.detach().clone() for n, p in model.named_parameters()},
               torch=torch.__version__, note="oracle/make_golden_lm.py: reference SimpleLMHeadModel, fp32, CPU")
    path = os.path.join(OUT, NAME)
    torch.save(out, path)
    print(NAME, os.path.getsize(path), "loss", float(loss), "params", sum(p.numel() for p in model.parameters()))


if __name__ == "__main__":
    main()
