"""HyenaDNALM pads batches of several odd-length sequences to a multiple of 64 positions (round 6, lm.HyenaDNALM._aligned_length): the reference
trainer's batches are (B, max_length - 1) (hg38_dataset.py:222, hg38_hyena.yaml:47-48).  Every operation of the model is causal or per-position,
so the padded run must give the unpadded run's logits, loss and gradients -- checked here under the CPU emulation of the kernels."""
import pytest
import torch

import hyena_dna_amd.lm as LM


def _model(L, d=64, n_layer=2, seed=0):
    torch.manual_seed(seed)
    layer = dict(l_max=L + 3, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10, lr=6e-4, wd=0.0, lr_pos_emb=0.0)
    return LM.HyenaDNALM(d_model=d, n_layer=n_layer, d_inner=4 * d, vocab_size=12, layer=layer, resid_dropout=0.0, embed_dropout=0.0,
                         pad_vocab_size_multiple=8, fused_dropout_add_ln=True, residual_in_fp32=True)


@pytest.mark.parametrize("B,L", [(3, 127), (2, 191), (1, 189)])
def test_padded_batch_equals_unpadded_batch(emu_backend, monkeypatch, B, L):
    model = _model(L)
    g = torch.Generator().manual_seed(L)
    ids = torch.randint(7, 11, (B, L), generator=g)
    tgt = torch.roll(ids, -1, 1)
    res = {}
    monkeypatch.setattr(LM, "_PAD_SINGLE_MIN", 100)             # (B = 1: a single sequence counts as long from here on)
    monkeypatch.setattr(LM, "_PAD_SINGLE_ROWS", 64)
    for pad in (True, False):
        monkeypatch.setattr(LM, "PAD_SEQUENCES", pad)
        assert model._aligned_length(ids) == (L + (-L) % 64 if pad else L)
        model.zero_grad(set_to_none=True)
        logits = model(ids)[0].logits
        assert logits.shape == (B, L, 16)
        loss = LM.token_cross_entropy(logits, tgt)
        loss.backward()
        res[pad] = (logits.detach().clone(), loss.item(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
    (la, lossa, ga), (lb, lossb, gb) = res[True], res[False]
    assert ((la - lb).norm() / lb.norm()).item() < 2e-5 and abs(lossa - lossb) < 1e-5 * abs(lossb)
    assert set(ga) == set(gb)
    for n in gb:
        e = ((ga[n] - gb[n]).norm() / gb[n].norm().clamp_min(1e-20)).item()
        assert e < 2e-4, (n, e)


def test_padding_only_where_it_applies(emu_backend, monkeypatch):
    monkeypatch.setattr(LM, "PAD_SEQUENCES", True)
    model = _model(127)                                                   # l_max = 130 admits 128
    assert model._aligned_length(torch.zeros(2, 127, dtype=torch.long)) == 128
    assert model._aligned_length(torch.zeros(1, 127, dtype=torch.long)) == 127        # one short sequence: its rows are aligned anyway
    monkeypatch.setattr(LM, "_PAD_SINGLE_MIN", 100)                                   # (one LONG sequence is padded: the odd weight-gradient products ...
    monkeypatch.setattr(LM, "_PAD_SINGLE_ROWS", 128)                                  #  ... where the padded length is a multiple of the slice grid)
    assert model._aligned_length(torch.zeros(1, 127, dtype=torch.long)) == 128
    monkeypatch.setattr(LM, "_PAD_SINGLE_ROWS", 4096)
    assert model._aligned_length(torch.zeros(1, 127, dtype=torch.long)) == 127
    monkeypatch.setattr(LM, "_PAD_SINGLE_MIN", 8192)
    big = _model(32767, d=64, n_layer=1)
    assert big._aligned_length(torch.zeros(1, 32767, dtype=torch.long)) == 32768 and big._aligned_length(torch.zeros(1, 32700, dtype=torch.long)) == 32700
    assert model._aligned_length(torch.zeros(4, 128, dtype=torch.long)) == 128
    assert model._aligned_length(torch.zeros(4, 33, dtype=torch.long)) == 33          # shorter than one 64-position tile
    tight = _model(124)                                                   # l_max = 127 < 128: the operator would truncate -> not padded
    assert tight._aligned_length(torch.zeros(2, 125, dtype=torch.long)) == 125
