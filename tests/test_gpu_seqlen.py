"""Sequence-length warm-up on a real MI355X (src/callbacks/seqlen_warmup_reload.py): the SAME model trains at
L = 1024 -> 32768 -> 160000 (both long-convolution plans), eagerly and through lm.GraphedTrainStep re-captured at every stage.
Asserted: per-length state is re-keyed (one twiddle table set per transform size, the workspace grows), results equal those
of a process that only ever saw that length (bitwise), going back to an earlier length reproduces its bits, and a hipGraph
captured at an early stage still replays correctly after later stages have outgrown the workspace it was captured on."""
import pytest
import torch

pytestmark = pytest.mark.gpu

STAGES = [(1024, 8), (32768, 2), (160000, 1)]


def _model(dev, l_max, D=128, n_layer=2):
    from hyena_dna_amd.lm import HyenaDNALM
    torch.manual_seed(11)
    layer = dict(l_max=l_max, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10, lr=6e-4, wd=0.0,
                 lr_pos_emb=0.0)
    return HyenaDNALM(d_model=D, n_layer=n_layer, d_inner=4 * D, vocab_size=12, layer=layer, resid_dropout=0.0, embed_dropout=0.0,
                      pad_vocab_size_multiple=8).to(dev)


def _loss_and_grads(model, ids):
    model.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = model.loss(ids, torch.roll(ids, -1, 1))
    loss.backward()
    return loss.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters()}


def test_eager_length_changes_rekey_state_and_match_fresh(gpu_lib):
    _lib = gpu_lib
    dev = torch.device("cuda", 0)
    model = _model(dev, STAGES[-1][0] + 2)
    g = torch.Generator(device=dev).manual_seed(5)
    batches = {s: torch.randint(7, 11, (s[1], s[0]), generator=g, device=dev) for s in STAGES}
    _lib._tables.clear()
    _lib._workspace.clear()
    order = STAGES + STAGES[:2]
    staged = [_loss_and_grads(model, batches[s]) for s in order]
    assert len({k[1] for k in _lib._tables if k[0] == dev.index}) == 3, list(_lib._tables)
    for i, j in ((0, 3), (1, 4)):                                   # back to an earlier length: same bits
        assert torch.equal(staged[i][0], staged[j][0])
        assert all(torch.equal(staged[i][1][n], staged[j][1][n]) for n in staged[i][1])
    for s, (loss, grads) in zip(STAGES, staged[:3]):                # a process that only ever saw this length
        torch.cuda.synchronize()
        _lib._tables.clear()
        _lib._workspace.clear()
        _lib._save_decision.clear()
        f_loss, f_grads = _loss_and_grads(model, batches[s])
        assert torch.equal(f_loss, loss), (s, float(f_loss), float(loss))
        for n in grads:
            assert torch.equal(f_grads[n], grads[n]), (s, n)


def test_graphed_step_recaptured_per_stage(gpu_lib):
    from hyena_dna_amd.lm import GraphedTrainStep
    dev = torch.device("cuda", 0)
    model = _model(dev, STAGES[-1][0] + 2)
    # lr = 0 and no weight decay: the captured optimizer step runs but leaves the weights alone, so every replay of every stage
    # is comparable with the eager loss on the same weights
    opt = torch.optim.AdamW(model.parameters(), lr=0.0, weight_decay=0.0, capturable=True)
    g = torch.Generator(device=dev).manual_seed(6)
    steps, first = [], []
    for L, B in STAGES:
        ids = torch.randint(7, 11, (B, L), generator=g, device=dev)
        eager_loss, _ = _loss_and_grads(model, ids)
        model.zero_grad(set_to_none=True)
        step = GraphedTrainStep(model, opt, ids, torch.roll(ids, -1, 1), warmup=1)      # re-capture at the new length
        loss = step().clone()
        assert torch.isfinite(loss) and abs(float(loss) - float(eager_loss)) <= 1e-3 * abs(float(eager_loss)), (L, float(loss), float(eager_loss))
        steps.append(step)
        first.append(loss)
    # the graphs of the earlier stages replay on their own (retired) workspaces after later stages outgrew them
    for step, want in zip(steps, first):
        for _ in range(2):
            got = step().clone()
            assert torch.allclose(got, want, rtol=1e-6, atol=0.0), (float(got), float(want))
    for p in model.parameters():
        assert bool(torch.isfinite(p).all())
