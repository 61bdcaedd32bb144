"""Host side above the C ABI -- the autograd function and the module mirrors -- on the CPU-emulated kernels,
against the oracle and the golden vectors minted from the reference's own classes.  No GPU needed."""
import os

import pytest
import torch

from oracle import hyena_oracle as O


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def test_product_refuses_cpu_tensors_and_missing_library(monkeypatch, tmp_path):
    """No fallback: CPU tensors raise, a missing .so raises -- never a silent torch path."""
    from hyena_dna_amd import _lib, fftconv
    u, k, D = torch.randn(1, 2, 16), torch.randn(2, 16), torch.randn(2)
    with pytest.raises(_lib.HyenaLibraryError, match="no CPU fallback"):
        fftconv.fftconv_func(u, k, D, gelu=False)
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib._backend, "path", str(tmp_path / "nope.so"), raising=False)
    with pytest.raises(_lib.HyenaLibraryError, match="not found"):
        _lib.lib()
    monkeypatch.setattr(_lib, "_lib", None)


def test_unsupported_options_raise(emu_backend):
    from hyena_dna_amd.fftconv import fftconv_func, fftconv_heads_ref
    u, k, D = torch.randn(1, 2, 16), torch.randn(2, 16), torch.randn(2)
    # since round 4 fftconv_func serves the H3-form options (tests/test_fftconv_options.py); what stays refused: the autograd class
    # called directly with them, head_dim > 1 without v / q, half of the H3 form, a dropout mask on top of the H3 form
    from hyena_dna_amd.fftconv import FFTConvFunc
    for kw in (dict(gelu=True), dict(gelu=False, dropout_mask=torch.ones(1, 2)), dict(gelu=False, output_hbl_layout=True)):
        with pytest.raises(NotImplementedError):
            FFTConvFunc.apply(u, k, D, kw.get("dropout_mask"), kw["gelu"], False, kw.get("output_hbl_layout", False))
    with pytest.raises(ValueError):
        fftconv_func(u, k, D, gelu=False, head_dim=8)
    with pytest.raises(ValueError):
        fftconv_func(u, k, D, gelu=False, q=u)
    with pytest.raises(NotImplementedError):
        fftconv_func(u, k, D, gelu=False, v=u, q=u, dropout_mask=torch.ones(1, 2))
    with pytest.raises(NotImplementedError):
        fftconv_heads_ref()
    with pytest.raises(ValueError):
        fftconv_func(torch.randn(1, 3, 16), k, D, gelu=False)


@pytest.mark.parametrize("five_d", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_autograd_matches_reference_autograd(emu_backend, five_d, dtype):
    from hyena_dna_amd.fftconv import fftconv_func
    g = torch.Generator().manual_seed(5)
    B, D, L = 2, 4, 300
    u = torch.randn(B, D, L, generator=g).to(dtype)
    k = torch.randn(D, L, generator=g) * torch.exp(-4 * torch.linspace(0, 1, L)) * 0.2
    bias = torch.randn(D, generator=g)
    dout = torch.randn(B, D, L, generator=g).to(dtype)

    def run(fn):
        u_ = u.clone().requires_grad_(True)
        k_ = k.clone().requires_grad_(True)
        b_ = bias.clone().requires_grad_(True)
        if five_d:      # exactly what HyenaOperator passes (hyena.py:396-423)
            out = fn(u_.reshape(B, 1, D, 1, L), k_, b_[None, :, None]).reshape(B, D, L)
        else:
            out = fn(u_, k_, b_)
        out.backward(dout)
        return out.detach(), u_.grad, k_.grad, b_.grad

    got = run(lambda a, b, c: fftconv_func(a, b, c, dropout_mask=None, gelu=False))
    ref = run(lambda a, b, c: O.fftconv_ref(a, b, c, None, gelu=False))
    tol = 2e-6 if dtype == torch.float32 else 1.2e-2
    for a, b in zip(got, ref):
        assert a.shape == b.shape and a.dtype == b.dtype
        assert _rel(a.float(), b.float()) < tol
    assert _rel(got[2], ref[2]) < 2e-6 and _rel(got[3], ref[3]) < 5e-6      # dk, dbias are fp32 on both sides


def test_multi_head_and_blocks_layout(emu_backend):
    """5-D input with h > 1 and z > 1: the filter broadcasts over b, h, z (hyena.py:77-78)."""
    from hyena_dna_amd.fftconv import fftconv_func
    g = torch.Generator().manual_seed(9)
    b, h, v, z, l = 2, 2, 3, 2, 40
    u = torch.randn(b, h, v, z, l, generator=g, requires_grad=True)
    k = torch.randn(v, l, generator=g, requires_grad=True)
    D = torch.randn(1, v, 1, generator=g, requires_grad=True)
    dout = torch.randn(b, h, v, z, l, generator=g)
    out = fftconv_func(u, k, D, gelu=False)
    out.backward(dout)
    got = (out.detach(), u.grad.clone(), k.grad.clone(), D.grad.clone())
    for t in (u, k, D):
        t.grad = None
    ref = O.fftconv_ref(u, k, D, None, gelu=False)
    ref.backward(dout)
    for a, r in zip(got, (ref.detach(), u.grad, k.grad, D.grad)):
        assert a.shape == r.shape and _rel(a, r) < 3e-6


def test_force_fp16_output(emu_backend):
    from hyena_dna_amd.fftconv import fftconv_func
    u, k, D = torch.randn(1, 2, 64), torch.randn(2, 64) * 0.1, torch.randn(2)
    assert fftconv_func(u, k, D, gelu=False, force_fp16_output=True).dtype == torch.float16
    assert fftconv_func(u.bfloat16(), k, D, gelu=False, force_fp16_output=True).dtype == torch.bfloat16


@pytest.mark.parametrize("name", ["d8l64", "d16l257", "d8l80_trunc"])
def test_operator_mirror_loads_reference_state_and_matches(emu_backend, golden_operator, name):
    """HyenaOperator mirror: load the REFERENCE module's state_dict, reproduce its output and every gradient."""
    from hyena_dna_amd.hyena import HyenaOperator
    c = golden_operator[name]
    op = HyenaOperator(d_model=c["d_model"], l_max=c["l_max"], order=2, filter_order=64, emb_dim=5,
                       short_filter_order=3, modulate=True, w=10, lr=6e-4, wd=0.0, lr_pos_emb=0.0,
                       layer_idx=0, device=None, dtype=None)         # stray kwargs as create_mixer_cls passes them
    missing, unexpected = op.load_state_dict(c["state_dict"], strict=True)
    assert not missing and not unexpected
    assert set(op.state_dict().keys()) == set(c["state_dict"].keys())
    assert torch.equal(op.filter_fn.filter(min(c["u"].shape[1], c["l_max"])), c["k"])
    u = c["u"].clone().requires_grad_(True)
    y = op(u)
    assert y.shape == c["y"].shape
    torch.testing.assert_close(y, c["y"], rtol=1e-5, atol=2e-6)
    y.backward(c["dy"])
    torch.testing.assert_close(u.grad, c["du"], rtol=1e-4, atol=2e-6)
    for n, p in op.named_parameters():
        torch.testing.assert_close(p.grad, c["grads"][n], rtol=2e-4, atol=2e-5, msg=lambda m, n=n: f"{n}: {m}")


def test_operator_mirror_autocast_bf16(emu_backend, golden_operator):
    from hyena_dna_amd.hyena import HyenaOperator
    c = golden_operator["d16l257"]
    op = HyenaOperator(d_model=c["d_model"], l_max=c["l_max"], order=2, filter_order=64, emb_dim=5,
                       short_filter_order=3, modulate=True, w=10, lr=6e-4, wd=0.0, lr_pos_emb=0.0)
    op.load_state_dict(c["state_dict"])
    with torch.autocast("cpu", dtype=torch.bfloat16):
        y = op(c["u"])
    assert y.dtype == c["y_autocast_bf16"].dtype
    assert _rel(y.float(), c["y_autocast_bf16"].float()) < 2e-2


def test_optim_tags_and_buffers_match_reference_contract():
    """_optim tags (src/utils/train.py:142-156) and buffer-vs-parameter split (lr == 0 -> buffer)."""
    from hyena_dna_amd.hyena import HyenaOperator
    op = HyenaOperator(d_model=8, l_max=34, order=2, filter_order=16, emb_dim=5, lr=6e-4, wd=0.1, lr_pos_emb=0.0, w=10)
    names = dict(op.named_parameters())
    bufs = dict(op.named_buffers())
    assert "filter_fn.pos_emb.z" in bufs and "filter_fn.pos_emb.t" in bufs and "filter_fn.modulation.deltas" in bufs
    assert names["filter_fn.implicit_filter.0.weight"]._optim == {"weight_decay": 0.1, "lr": 6e-4}
    assert names["filter_fn.implicit_filter.1.freq"]._optim == {"weight_decay": 0.1, "lr": 6e-4}
    assert not hasattr(names["in_proj.weight"], "_optim")
    op2 = HyenaOperator(d_model=8, l_max=34, emb_dim=5, lr_pos_emb=1e-5)
    assert dict(op2.named_parameters())["filter_fn.pos_emb.z"]._optim == {"lr": 1e-5, "weight_decay": 0.0}
    assert op.d_output == 8
    # one shared Sin instance: the three `freq` keys alias one tensor
    sd = op.state_dict()
    assert sd["filter_fn.implicit_filter.1.freq"].data_ptr() == sd["filter_fn.implicit_filter.5.freq"].data_ptr()
    with pytest.raises(AssertionError):
        HyenaOperator(d_model=8, l_max=34, order=1)
    with pytest.raises(ImportError):
        HyenaOperator(d_model=8, l_max=34, fused_bias_fc=True)


def test_tail_product_masked_slice_branch_on_host(monkeypatch):
    """projection._tail_product's GPU branch -- the last 256 positions as one 16-bit slice whose already-counted columns are zeroed -- taken for host
    tensors (projection._MASKED_TAIL_ON_HOST, tests only): equals the plain product of the leftover rows, through all three weight-gradient callers'
    operand layouts (that indexing was reachable on a GPU only)"""
    import hyena_dna_amd.projection as P
    g = torch.Generator().manual_seed(5)
    for n, done in ((70001, 69888), (999, 768), (256, 1), (300, 299), (513, 513)):
        a = torch.randn(24, n, generator=g).to(torch.bfloat16)                    # (C, n) as dy2.t() / a channel-major matrix
        b = torch.randn(n, 16, generator=g).to(torch.bfloat16)
        want = torch.mm(a[:, done:].float(), b[done:].float()) if done < n else None
        monkeypatch.setattr(P, "_MASKED_TAIL_ON_HOST", False)
        plain = P._tail_product(a, b, done)
        monkeypatch.setattr(P, "_MASKED_TAIL_ON_HOST", True)
        for av, bv in ((a, b), (a.t().contiguous().t(), b), (a, b.t().contiguous().t())):       # row- and column-major views of both operands
            got = P._tail_product(av, bv, done)
            if want is None:
                assert got is None and plain is None
                continue
            assert torch.allclose(got, want, rtol=1e-5, atol=1e-4) and torch.allclose(plain, want, rtol=1e-5, atol=1e-4), (n, done)
        assert a[:, n - min(n, 256):].abs().sum() > 0                                          # (the masking works on a copy)
    # and through the whole split: an odd row count whose plan leaves a tail
    monkeypatch.setattr(P, "_MASKED_TAIL_ON_HOST", True)
    dy2 = torch.randn(70001, 24, generator=g).to(torch.bfloat16)
    x2 = torch.randn(70001, 16, generator=g).to(torch.bfloat16)
    assert P.split_plan(70001, 24 * 16)[1] < 70001
    ref = torch.mm(dy2.t().double(), x2.double())
    got = P.split_k_weight_grad(dy2, x2).double()
    assert ((got - ref).norm() / ref.norm()).item() < 1e-6


def test_leftover_rows_as_one_padded_batched_product(monkeypatch):
    """projection._leftover_product (round 6): what split_plan leaves behind its first level -- second level + tail -- copied into zero-padded scratch
    operands and multiplied as ONE batch of 256-row slices; through the three weight-gradient callers and their operand layouts, against the float64
    products, and equal to the split form of round 5 to fp32 rounding"""
    import hyena_dna_amd.projection as P
    from hyena_dna_amd import _lib
    monkeypatch.setattr(P, "_MASKED_TAIL_ON_HOST", True)
    g = torch.Generator().manual_seed(11)
    for B, L in ((1, 70001), (8, 4097), (3, 12001)):
        rows = B * L
        C, K, N = 24, 16, 8
        d = _lib.empty_cm(C, B, L, torch.bfloat16, torch.device("cpu"))
        d.copy_(torch.randn(C, B, L, generator=g).to(torch.bfloat16))
        z = _lib.empty_cm(K, B, L, torch.bfloat16, torch.device("cpu"))
        z.copy_(torch.randn(K, B, L, generator=g).to(torch.bfloat16))
        x2 = torch.randn(rows, K, generator=g).to(torch.bfloat16)
        dy2 = torch.randn(rows, N, generator=g).to(torch.bfloat16)
        assert P.split_plan(rows, C * K)[0][0][1] * P.split_plan(rows, C * K)[0][0][2] < rows          # (there IS a leftover)
        want = {"cm_pm": d.reshape(C, rows).double() @ x2.double(), "pm_cm": dy2.t().double() @ z.reshape(K, rows).t().double(),
                "pm_pm": dy2.t().double() @ x2.double()}
        res = {}
        for merged in (True, False):
            monkeypatch.setattr(P, "MERGE_LEFTOVER", merged)
            res[merged] = {"cm_pm": P.wgrad_cm_pm(d, x2), "pm_cm": P.wgrad_pm_cm(dy2, z), "pm_pm": P.split_k_weight_grad(dy2, x2)}
        for name, ref in want.items():
            for merged in (True, False):
                got = res[merged][name]
                assert got.dtype == torch.float32 and ((got.double() - ref).norm() / ref.norm()).item() < 1e-6, (name, merged, B, L)
            assert torch.allclose(res[True][name], res[False][name], rtol=1e-5, atol=1e-3)
    # fp32 operands: a plain product of the leftover rows
    a = torch.randn(24, 5000, generator=g)
    b = torch.randn(5000, 16, generator=g)
    assert torch.allclose(P._leftover_product(a, b, 4608), a[:, 4608:] @ b[4608:], rtol=1e-5, atol=1e-5) and P._leftover_product(a, b, 5000) is None


def test_split_k_linear_gradients_match_linear():
    """hyena_dna_amd/projection.py: the slice-batched weight gradient equals autograd's dy^T x (hyena.py:391,440)"""
    from hyena_dna_amd.projection import SplitKLinearFunc, split_count
    assert split_count(1 << 20) == 64 and split_count(32768) == 8 and split_count(160000) == 32 and split_count(8191) == 2 and split_count(8000) == 1
    # the lengths the reference dataset really yields are max_length - 1 (hg38_dataset.py:220): odd row counts split too
    assert split_count(999999) == 64 and split_count(449999) == 64 and split_count(159999) == 32 and split_count(2 * 32767) == 16 and split_count(8 * 32767) == 64
    # round 5: a row count that its slice count does not divide is cut into 256-row-aligned slices on two levels + a tail of < 256 rows
    from hyena_dna_amd.projection import split_plan
    assert split_plan(1 << 20) == ([(0, 64, 16384)], 1 << 20)                                   # divisible: one level, as before
    assert split_plan(1 << 20, 256 * 256) == ([(0, 256, 4096)], 1 << 20) and split_count(32768, 256 * 256) == 8     # a one-tile gradient: more slices
    # round 6: equal slices of an ODD number of rows (10^6 = 64 x 15625) are the slow case themselves: cut like a count the slice count does not divide
    assert split_plan(1000000) == ([(0, 64, 15616), (999424, 2, 256)], 999936) and split_plan(64 * 4092) == ([(0, 64, 4092)], 64 * 4092)
    for rows in (1048575, 999999, 449999, 2 * 159999, 8 * 32767, 32767, 70001, 1000000):
        levels, done = split_plan(rows)
        pos = 0
        for p0, s_, q in levels:
            assert p0 == pos and q % 64 == 0 and s_ >= 1
            pos += s_ * q
        assert pos == done and 0 <= rows - done < 256
    for dt, tol, rows in ((torch.float32, 1e-6, 8192), (torch.bfloat16, 1e-2, 8192), (torch.float32, 1e-6, 16383), (torch.bfloat16, 1e-2, 9999),
                          (torch.float32, 1e-6, 35001)):
        g = torch.Generator().manual_seed(0)
        x = torch.randn(2, rows, 16, generator=g).to(dt).requires_grad_()
        w = (torch.randn(24, 16, generator=g) * 0.1).to(dt).requires_grad_()
        b = torch.randn(24, generator=g).to(dt).requires_grad_()
        y = SplitKLinearFunc.apply(x, w, b)
        dy = torch.randn(y.shape, generator=g).to(dt)
        y.backward(dy)
        got = [t.grad.clone() for t in (x, w, b)]
        for t in (x, w, b):
            t.grad = None
        ref = torch.nn.functional.linear(x, w, b)
        assert torch.equal(y, ref)
        ref.backward(dy)
        for a, t in zip(got, (x, w, b)):
            assert a.dtype == t.grad.dtype
            assert ((a.float() - t.grad.float()).norm() / t.grad.float().norm()).item() < tol


def test_empty_batch_and_oversize_length(emu_backend):
    """edge cases at the op seam: an empty batch behaves like torch.fft's path (empty output, zero parameter gradients);
    a length beyond HYENA_MAX_L is refused loudly instead of silently taking another path"""
    from hyena_dna_amd.fftconv import fftconv_func
    from hyena_dna_amd._lib import HyenaLibraryError
    u = torch.zeros(0, 3, 50, requires_grad=True)
    k = torch.randn(3, 50, requires_grad=True)
    bias = torch.randn(3, requires_grad=True)
    y = fftconv_func(u, k, bias, dropout_mask=None, gelu=False)
    assert y.shape == (0, 3, 50)
    y.sum().backward()
    assert torch.count_nonzero(k.grad) == 0 and torch.count_nonzero(bias.grad) == 0 and u.grad.shape == u.shape
    with pytest.raises(HyenaLibraryError):
        emu_backend.fftconv_fwd(torch.zeros(1, 1, (1 << 20) + 1), torch.zeros(1, (1 << 20) + 1), None)
    # the fused operator path too: an empty batch returns an empty tensor, parameter gradients are zeros
    from hyena_dna_amd.hyena import HyenaOperator
    op = HyenaOperator(d_model=64, l_max=102, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10)
    x = torch.randn(0, 100, 64, requires_grad=True)
    y = op(x)
    assert y.shape == (0, 100, 64)
    y.sum().backward()
    assert x.grad.shape == x.shape and all(p.grad is None or torch.count_nonzero(p.grad) == 0 for p in op.parameters())


def test_tiny_lm_trains_on_the_emulated_kernels(emu_backend):
    """all autograd Functions together (fused filter, mixer shell, long conv, add+LayerNorm, tokenizer): the loss of a
    2-layer stack on periodic DNA falls"""
    from tests._tiny_lm import train
    losses = train("cpu", steps=8, d=64, L=70, B=2, n_layer=2)
    assert all(l == l for l in losses)                       # finite
    assert losses[-1] < 0.8 * losses[0], losses


def test_block_glue_refuses_cpu_tensors_without_the_test_double():
    """no silent CPU path for the add + LayerNorm either (the emulator backend is only ever installed by tests)"""
    from hyena_dna_amd.block import dropout_add_layer_norm
    from hyena_dna_amd._lib import HyenaLibraryError
    with pytest.raises(HyenaLibraryError):
        dropout_add_layer_norm(torch.randn(2, 3, 64), None, torch.ones(64), torch.zeros(64), 0.0, 1e-5, residual_in_fp32=True)


def test_graphed_train_step_refuses_what_it_cannot_capture():
    """lm.GraphedTrainStep fails loudly, before touching the device, on a CPU batch and on a non-capturable optimizer.  The ROCm graph
    knob is the caller's business since round 4 (the test session sets it in conftest.py, ahead of every import)."""
    import os

    import hyena_dna_amd
    from hyena_dna_amd.lm import GraphedTrainStep
    assert os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0" and isinstance(hyena_dna_amd.GRAPH_SAFE, bool)
    lin = torch.nn.Linear(4, 4)
    ids = torch.zeros(1, 8, dtype=torch.long)
    with pytest.raises(RuntimeError, match="ROCm device"):
        GraphedTrainStep(lin, torch.optim.AdamW(lin.parameters(), lr=1e-3), ids, ids)


def test_workspace_growth_frees_unless_captured_and_save_decision_is_cached(emu_backend, monkeypatch):
    """(a) an outgrown workspace is dropped, not retired forever, unless a hipGraph capture was handed the buffer;
    (b) a forward that a backward will follow takes the backward's (larger) workspace right away; (c) the two-level plan's
    keep-the-spectra decision is made once per (device, B, D, L), not re-derived from the allocator's state every step."""
    _lib = emu_backend
    monkeypatch.setattr(_lib, "_retired", [])
    monkeypatch.setattr(_lib, "_captured", set())
    dev = torch.device("cpu")
    w1, _ = _lib.workspace_for(dev, 1000)
    w2, _ = _lib.workspace_for(dev, 5000)
    assert w2.numel() >= 5000 and _lib._retired == []
    monkeypatch.setattr(_lib._backend, "capturing", lambda: True, raising=False)
    w3, _ = _lib.workspace_for(dev, 100)                      # handed out during a capture
    assert w3 is w2
    monkeypatch.setattr(_lib._backend, "capturing", lambda: False, raising=False)
    w4, _ = _lib.workspace_for(dev, 50000)
    assert [r for _, r in _lib._retired] == [w2] and _lib._retired[0][0][1] == _lib._backend.stream(dev) and w4.numel() >= 50000
    # (b)
    B, D, L = 2, 4, 40000                                       # two-level plan
    u, k = torch.randn(B, D, L), torch.randn(D, L) * 0.01
    monkeypatch.setattr(_lib, "_workspace", {})
    _lib.fftconv_fwd(u, k, None, grad=True)
    need_bwd = _lib.lib().hyena_fftconv_workspace_bytes(B, D, L, 1, 0)
    assert next(iter(_lib._workspace.values())).numel() >= need_bwd
    # (c)
    monkeypatch.setattr(_lib, "_save_decision", {})
    calls = []

    def free(device=None):
        calls.append(1)
        return 1 << 40
    monkeypatch.setattr(_lib._backend, "free_memory", free, raising=False)
    assert _lib.save_spectra_default(B, D, L, device=dev) and _lib.save_spectra_default(B, D, L, device=dev)
    assert len(calls) == 1


def test_lm_checkpoint_flags_wrap_like_the_reference_and_unknown_keywords_raise():
    """checkpoint_mixer / checkpoint_mlp wrap the sub-modules under `.layer` (long_conv_lm.py:39-45, 196-199: the
    state-dict keys move), everything the model has no counterpart for raises instead of being swallowed."""
    from hyena_dna_amd.lm import CheckpointedModule, HyenaDNALM
    layer = dict(l_max=66, order=2, filter_order=16, emb_dim=5, short_filter_order=3, modulate=True, w=10, lr=6e-4, wd=0.0, lr_pos_emb=0.0)
    kw = dict(d_model=16, n_layer=2, d_inner=32, vocab_size=12, layer=layer, pad_vocab_size_multiple=8)
    m = HyenaDNALM(checkpoint_mixer=True, checkpoint_mlp=True, **kw)
    assert isinstance(m.backbone.layers[0].mixer, CheckpointedModule) and isinstance(m.backbone.layers[1].mlp, CheckpointedModule)
    keys = set(m.state_dict())
    assert "backbone.layers.0.mixer.layer.in_proj.weight" in keys and "backbone.layers.1.mlp.layer.fc1.weight" in keys
    plain = HyenaDNALM(**kw)
    assert "backbone.layers.0.mixer.in_proj.weight" in set(plain.state_dict())
    HyenaDNALM(attn_layer_idx=None, fused_mlp=False, process_group=None, **kw)            # reference keywords at "off": fine
    with pytest.raises(NotImplementedError):
        HyenaDNALM(attn_layer_idx=[1], **kw)
    with pytest.raises(NotImplementedError):
        HyenaDNALM(fused_mlp=True, **kw)
    with pytest.raises(TypeError):
        HyenaDNALM(no_such_option=1, **kw)


@pytest.mark.parametrize("L", [64, 97])
def test_k_rev_and_bidirectional_match_the_reference_definition(emu_backend, L):
    """VERDICT r2 missing 7: `k_rev` (src/ops/fftconv.py:64-66) and `bidirectional` (hyena.py:67-73) through the HIP kernels (flips /
    a delay around the causal convolution) against the oracle's restatement of the reference's circular 2L-point definition, values
    and gradients; and the operator builds and runs with bidirectional=True (the reference's README "Experimental" configuration)."""
    from hyena_dna_amd.fftconv import fftconv_func, fftconv_ref
    from oracle import hyena_oracle as O
    g = torch.Generator().manual_seed(L)
    B, D = 2, 3
    mk = lambda *s: torch.randn(*s, generator=g)          # noqa: E731
    u, k, kr, bias, dout = mk(B, D, L), mk(D, L) * 0.2, mk(D, L) * 0.2, mk(D), mk(B, D, L)
    for kind in ("k_rev", "bidirectional"):
        leaves = [t.clone().requires_grad_(True) for t in (u, k, kr, bias)]
        refs = [t.clone().requires_grad_(True) for t in (u, k, kr, bias)]
        kw = dict(k_rev=leaves[2] if kind != "bidirectional" else None)
        kwr = dict(k_rev=refs[2] if kind != "bidirectional" else None)
        if kind == "k_rev":
            y = fftconv_func(leaves[0], leaves[1], leaves[3], gelu=False, **kw)
        else:
            y = fftconv_ref(leaves[0], leaves[1], leaves[3], gelu=False, bidirectional=True, **kw)
        yr = O.fftconv_ref(refs[0], refs[1], refs[3], None, gelu=False, bidirectional=(kind != "k_rev"), **kwr)
        y.backward(dout)
        yr.backward(dout)
        assert _rel(y, yr) < 2e-6, (kind, _rel(y, yr))
        for a, r, n in zip(leaves, refs, ("du", "dk", "dk_rev", "dbias")):
            if r.grad is None:
                assert a.grad is None or torch.count_nonzero(a.grad) == 0, (kind, n)
            else:
                assert _rel(a.grad, r.grad) < 5e-6, (kind, n, _rel(a.grad, r.grad))
    from hyena_dna_amd.hyena import HyenaOperator
    torch.manual_seed(0)
    op = HyenaOperator(d_model=8, l_max=L, order=2, filter_order=16, emb_dim=3, bidirectional=True)
    x = torch.randn(2, L, 8, requires_grad=True)
    yo = op(x)
    sd = {n: v.detach() for n, v in op.state_dict().items()}
    ref = O.hyena_operator(sd, x.detach(), l_max=L, conv_fn=lambda v, kk, bb, m, gelu: O.fftconv_ref(v, kk, bb, m, gelu=gelu, bidirectional=True))
    assert _rel(yo, ref) < 5e-6
    yo.sum().backward()
    assert x.grad is not None and torch.isfinite(x.grad).all()


def test_advice_r3_host_side_fixes(tmp_path, monkeypatch):
    """unset hg38 paths fall back to the reference's default location and fail with an error that names the overrides;
    HyenaDNALM applies the `device` / `dtype` factory keywords the reference's modules take; the cached keep-the-spectra decisions can be
    dropped per device; a released capture stream gives its workspace back."""
    import os
    from hyena_dna_amd import _lib, runner
    from hyena_dna_amd.lm import HyenaDNALM
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = runner.compose(os.path.join(root, "tests", "golden", "hg38_hyena_composed.json"))
    monkeypatch.chdir(tmp_path)
    with pytest.raises(FileNotFoundError, match="dataset.bed_file"):
        runner.build_dataset(cfg)
    m = HyenaDNALM(d_model=64, n_layer=1, d_inner=256, vocab_size=12, layer=dict(l_max=66, order=2, filter_order=64, emb_dim=5),
                   dtype=torch.bfloat16, device="cpu")
    assert all(p.dtype == torch.bfloat16 for p in m.parameters())
    monkeypatch.setattr(_lib, "_save_decision", {(0, 1, 2, 3): True, (1, 1, 2, 3): False})
    _lib.reset_save_decisions(torch.device("cuda", 0))
    assert _lib._save_decision == {(1, 1, 2, 3): False}
    _lib.reset_save_decisions()
    assert _lib._save_decision == {}
    w = torch.empty(16, dtype=torch.uint8)
    monkeypatch.setattr(_lib, "_workspace", {(0, 77): w})
    monkeypatch.setattr(_lib, "_captured", {(0, 77)})
    other = torch.empty(8, dtype=torch.uint8)
    monkeypatch.setattr(_lib, "_retired", [((0, 77), torch.empty(4, dtype=torch.uint8)), ((0, 78), other)])     # outgrown buffers, by owner
    assert _lib.release_stream_state(torch.device("cuda", 0), 77) and not _lib._workspace and not _lib._captured
    assert len(_lib._retired) == 1 and _lib._retired[0][0] == (0, 78) and _lib._retired[0][1] is other             # this stream's outgrown buffers go, another stream's stay
    assert not _lib.release_stream_state(torch.device("cuda", 0), 77)


def test_importing_the_package_leaves_the_process_environment_alone():
    """`import hyena_dna_amd` used to set DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 for the whole process.  It is opt-in now:
    prepare_graph_runtime() sets it (only while the HIP runtime is uninitialised, only if the user has not chosen a value)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os; os.environ.pop('DEBUG_CLR_GRAPH_PACKET_CAPTURE', None)\n"
            "import hyena_dna_amd as h\n"
            "assert 'DEBUG_CLR_GRAPH_PACKET_CAPTURE' not in os.environ and h.GRAPH_SAFE is False\n"
            "assert h.prepare_graph_runtime() is True and os.environ['DEBUG_CLR_GRAPH_PACKET_CAPTURE'] == '0' and h.GRAPH_SAFE is True\n"
            "os.environ['DEBUG_CLR_GRAPH_PACKET_CAPTURE'] = '1'; h.GRAPH_SAFE = False\n"
            "assert h.prepare_graph_runtime() is False and os.environ['DEBUG_CLR_GRAPH_PACKET_CAPTURE'] == '1'\n"
            "print('ENV_OK')")
    env = {k: v for k, v in os.environ.items() if k != "DEBUG_CLR_GRAPH_PACKET_CAPTURE"}
    p = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "ENV_OK" in p.stdout, (p.stdout, p.stderr[-1500:])


def test_no_grad_forward_keeps_nothing_for_a_backward(emu_backend, monkeypatch):
    """ctx.needs_input_grad ignores torch.no_grad(): the wrappers pass the caller's grad mode on (hyena_dna_amd._gradmode), so serving a
    model whose parameters still require grad neither saves the column spectra nor asks the filter for its backward state."""
    import hyena_dna_amd.fftconv as FC
    from hyena_dna_amd.hyena import HyenaOperator
    seen = {"save": [], "filter_save": []}
    real_fwd, real_filter = emu_backend.fftconv_fwd, emu_backend.filter_fwd
    monkeypatch.setattr(emu_backend, "fftconv_fwd", lambda *a, **k: (seen["save"].append((k.get("save", False), k.get("grad"))), real_fwd(*a, **k))[1])
    monkeypatch.setattr(emu_backend, "filter_fwd", lambda *a, **k: (seen["filter_save"].append(k.get("save", False)), real_filter(*a, **k))[1])
    monkeypatch.setattr(emu_backend, "save_spectra_default", lambda *a, **k: True)
    torch.manual_seed(0)
    op = HyenaOperator(d_model=64, l_max=96, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10)
    u = torch.randn(2, 96, 64)
    y_train = op(u)
    assert seen["save"] == [(True, None)] and seen["filter_save"] == [True]
    seen["save"].clear(), seen["filter_save"].clear()
    with torch.no_grad():
        y_serve = op(u)
    assert seen["save"] == [(False, False)] and seen["filter_save"] == [False]
    assert torch.equal(y_train.detach(), y_serve)
    # the H3-form entry point alike
    seen["save"].clear()
    k = torch.randn(64, 96, requires_grad=True)
    with torch.no_grad():
        FC.fftconv_func(torch.randn(2, 64, 96), k, torch.randn(64), None, False)
    assert seen["save"] == [(False, False)]


def test_token_cross_entropy_is_f_cross_entropy():
    """lm.token_cross_entropy (log-softmax + gather + masked mean) == F.cross_entropy with ignore_index: value and gradient"""
    from hyena_dna_amd.lm import token_cross_entropy
    torch.manual_seed(0)
    logits = torch.randn(3, 50, 16, requires_grad=True)
    tgt = torch.randint(0, 16, (3, 50))
    tgt[0, :7] = -100
    ref = torch.nn.functional.cross_entropy(logits.float().reshape(-1, 16), tgt.reshape(-1), ignore_index=-100)
    g_ref, = torch.autograd.grad(ref, logits)
    got = token_cross_entropy(logits, tgt)
    g_got, = torch.autograd.grad(got, logits)
    assert torch.allclose(got, ref, rtol=1e-6, atol=1e-7) and torch.allclose(g_got, g_ref, rtol=1e-5, atol=1e-8)
    lb = logits.detach().to(torch.bfloat16)                    # 16-bit logits are evaluated in fp32, like the reference's metric
    assert torch.allclose(token_cross_entropy(lb, tgt), torch.nn.functional.cross_entropy(lb.float().reshape(-1, 16), tgt.reshape(-1)), rtol=1e-6)
