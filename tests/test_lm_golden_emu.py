"""HyenaDNALM against the golden minted from the reference's SimpleLMHeadModel (oracle/make_golden_lm.py; simple_lm.py:26-305),
kernels under tests/hipemu: logits, loss and every parameter gradient of the 2-layer d_model-128 stack at L = 4096.  The same
fixture pins the gfx950 binary in tests/test_gpu_contract.py."""
import os

import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lm_simple_d128_l4096.pt")


def _rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def test_lm_matches_reference_simple_lm_golden(emu_backend):
    from hyena_dna_amd.lm import HyenaDNALM
    c = torch.load(GOLDEN, weights_only=False)
    model = HyenaDNALM(layer=dict(c["layer"]), fused_dropout_add_ln=True, **c["cfg"])
    model.load_state_dict(c["state_dict"], strict=True)           # the reference's state-dict names, nothing missing or extra
    logits = model(c["ids"])[0].logits
    loss = torch.nn.functional.cross_entropy(logits.float().reshape(-1, logits.shape[-1]), c["targets"].reshape(-1))
    loss.backward()
    assert _rel(logits, c["logits"]) < 5e-6
    assert abs(loss.item() - c["loss"]) < 1e-6 * abs(c["loss"]) + 1e-7
    grads = {n: p.grad for n, p in model.named_parameters()}
    assert set(grads) == set(c["grads"])
    for n, g in c["grads"].items():
        assert _rel(grads[n], g) < 2e-5, (n, _rel(grads[n], g))


def test_lm_with_embedding_dropout_on_the_fused_pass(emu_backend):
    """HyenaDNALM in training mode with embed_dropout > 0: the first block's dropout runs inside the fused add + LayerNorm pass (block.py);
    the loss is reproducible under torch.manual_seed, differs between draws, eval mode applies none, and every parameter gets a gradient."""
    import torch
    from hyena_dna_amd.lm import HyenaDNALM
    L, D = 64, 64
    layer = dict(l_max=L + 2, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10)
    torch.manual_seed(0)
    m = HyenaDNALM(d_model=D, n_layer=2, d_inner=4 * D, vocab_size=12, layer=layer, resid_dropout=0.0, embed_dropout=0.3,
                   pad_vocab_size_multiple=8, fused_dropout_add_ln=True, residual_in_fp32=True)
    ids = torch.randint(7, 11, (2, L))
    tgt = torch.roll(ids, -1, 1)
    m.train()
    torch.manual_seed(11)
    a = m.loss(ids, tgt)
    b = m.loss(ids, tgt)
    torch.manual_seed(11)
    a2 = m.loss(ids, tgt)
    assert torch.equal(a, a2) and not torch.equal(a, b)
    a2.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters() if p.requires_grad)
    m.eval()
    with torch.no_grad():
        e1, e2 = m.loss(ids, tgt), m.loss(ids, tgt)
    assert torch.equal(e1, e2)
