"""Sequence-length warm-up (src/callbacks/seqlen_warmup_reload.py: the data loaders are rebuilt with a new max_length / batch size
at stage boundaries while the SAME model -- l_max of the final stage -- keeps training): every length-keyed piece of state in
this package (twiddle tables per transform size and plan, the per-stream workspace, the keep-the-spectra decision, the fused
filter's slice of the l_max-long position embedding) must be re-keyed when L changes mid-run, and the results must equal those of
a process that only ever saw that one length.  CPU half (kernels under tests/hipemu); the GPU half incl. hipGraph re-capture is
in tests/test_gpu_seqlen.py."""
import torch


def _model(l_max, D=64, n_layer=1):
    from hyena_dna_amd.lm import HyenaDNALM
    torch.manual_seed(11)
    layer = dict(l_max=l_max, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10, lr=6e-4, wd=0.0,
                 lr_pos_emb=0.0)
    return HyenaDNALM(d_model=D, n_layer=n_layer, d_inner=2 * D, vocab_size=12, layer=layer, resid_dropout=0.0, embed_dropout=0.0,
                      pad_vocab_size_multiple=8)


def _loss_and_grads(model, ids):
    model.zero_grad(set_to_none=True)
    loss = model.loss(ids, torch.roll(ids, -1, 1))
    loss.backward()
    return loss.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters()}


def test_length_changes_mid_run_rekey_tables_and_workspace(emu_backend):
    _lib = emu_backend
    stages = [(64, 4), (1100, 2), (33000, 1), (64, 4), (1100, 2)]           # workspace-free plan twice, two-level plan, and back
    model = _model(l_max=33002)
    g = torch.Generator().manual_seed(5)
    batches = {(L, B): torch.randint(7, 11, (B, L), generator=g) for L, B in set(stages)}
    staged = [_loss_and_grads(model, batches[s]) for s in stages]
    sizes = {k[1] for k in _lib._tables}
    assert len(_lib._tables) == 3 and len(sizes) == 3, _lib._tables.keys()   # one table set per transform size / plan
    # going back to a length seen before gives the same bits (tables and workspace reused, nothing stale)
    for i, j in ((0, 3), (1, 4)):
        assert torch.equal(staged[i][0], staged[j][0])
        assert all(torch.equal(staged[i][1][n], staged[j][1][n]) for n in staged[i][1])
    # a "fresh process" per length: all cached state dropped before each
    for (L, B), (loss, grads) in zip(stages[:3], staged[:3]):
        _lib._tables.clear()
        _lib._workspace.clear()
        _lib._save_decision.clear()
        f_loss, f_grads = _loss_and_grads(model, batches[(L, B)])
        assert torch.equal(f_loss, loss), (L, float(f_loss), float(loss))
        for n in grads:
            assert torch.equal(f_grads[n], grads[n]), (L, n)
