"""The matrix-core input projection with the front of the shell in its epilogue (csrc/proj_kernels.h, include/hyena_proj.h) under
tests/hipemu: xT against the fp32 product of the same 16-bit operands, vg BIT-IDENTICAL to cm_pre_fwd applied to that xT (the
kernel it replaces), ragged position counts (tiles crossing sequence boundaries, B Lx not a multiple of 64), truncation
(Lc < Lx), both widths and both 16-bit types."""
import pytest
import torch


@pytest.fixture(params=[1, 2])
def outproj_gen(request, emu_backend):
    """both generations of the out_proj forward kernel (include/hyena_proj.h, hyena_proj_kernel_generation): round 4's and round 6's"""
    prev = emu_backend.proj_kernel_generation(0)
    assert emu_backend.proj_kernel_generation(0, request.param) == request.param
    yield request.param
    emu_backend.proj_kernel_generation(0, prev)


@pytest.fixture(params=[1, 2])
def inproj_gen(request, emu_backend):
    """both generations of the in_proj forward kernel: rounds 3 / 4's and round 6's (default)"""
    prev = emu_backend.proj_kernel_generation(1)
    assert emu_backend.proj_kernel_generation(1, request.param) == request.param
    yield request.param
    emu_backend.proj_kernel_generation(1, prev)


def _case(B, Lx, D, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    u = torch.randn(B, Lx, D, generator=g).to(dtype)
    W = (torch.randn(3 * D, D, generator=g) / D ** 0.5).to(dtype)
    bin_ = torch.randn(3 * D, generator=g) * 0.3
    w = torch.randn(3 * D, 3, generator=g) * 0.5
    b = torch.randn(3 * D, generator=g) * 0.2
    return u, W, bin_, w, b


@pytest.mark.parametrize("B,Lx,Lc,D,dtype", [(1, 64, 64, 128, torch.bfloat16), (2, 100, 100, 128, torch.bfloat16), (3, 77, 70, 128, torch.float16),
                                             (1, 700, 700, 256, torch.bfloat16), (2, 333, 300, 256, torch.float16), (5, 9, 9, 128, torch.bfloat16),
                                             (1, 1500, 1500, 128, torch.bfloat16), (4, 1023, 1023, 128, torch.bfloat16), (3, 130, 100, 256, torch.bfloat16),
                                             (1, 2111, 2111, 256, torch.float16)])
def test_inproj_pre_fwd_vs_gemm_and_cm_pre(emu_backend, inproj_gen, B, Lx, Lc, D, dtype):
    _lib = emu_backend
    u, W, bin_, w, b = _case(B, Lx, D, dtype, seed=Lx + D)
    assert _lib.proj_supported(B, Lx, D, dtype)
    xT, vg = _lib.inproj_pre_fwd(u, W, bin_, w, b, Lc)
    ref = torch.mm(W.float(), u.float().reshape(B * Lx, D).t()).view(3 * D, B, Lx)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert ((xT.float() - ref).abs() <= eps * ref.abs() + 1e-6).all()                       # one rounding of the fp32 sum
    vg_ref = _lib.cm_pre_fwd(xT, bin_, w, b, Lc)
    assert torch.equal(vg, vg_ref)
    # no bias
    xT0, vg0 = _lib.inproj_pre_fwd(u, W, None, w, b, Lc)
    assert torch.equal(xT0, xT) and torch.equal(vg0, _lib.cm_pre_fwd(xT, None, w, b, Lc))


def test_proj_supported_shapes(emu_backend):
    _lib = emu_backend
    assert not _lib.proj_supported(1, 100, 64, torch.bfloat16) and not _lib.proj_supported(1, 100, 128, torch.float32)
    assert not _lib.proj_supported(1, 4, 128, torch.bfloat16) and _lib.proj_supported(8, 32768, 256, torch.float16)


def test_operator_with_and_without_the_mfma_projection(emu_backend, inproj_gen, monkeypatch):
    """HyenaOperator (d_model 128, bf16 tensors): the matrix-core in_proj + epilogue against the library GEMM + cm_pre_fwd path --
    the same vg bits for the same xT, so the two runs differ only by the GEMMs' own rounding of xT."""
    import hyena_dna_amd.projection as P
    from hyena_dna_amd.hyena import HyenaOperator
    torch.manual_seed(3)
    B, L, D = 2, 150, 128
    op = HyenaOperator(d_model=D, l_max=L + 2, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10,
                       lr=6e-4, wd=0.0, lr_pos_emb=0.0)
    with torch.no_grad():
        op.in_proj.bias.normal_(0, 0.1)
    op = op.to(torch.bfloat16)
    for n, buf in op.named_buffers():              # the filter's own tensors stay fp32 in training (autocast); keep them so here
        pass
    u0 = torch.randn(B, L, D).to(torch.bfloat16)
    dy = torch.randn(B, L, D).to(torch.bfloat16)
    res = []
    calls = []
    real = emu_backend.inproj_pre_fwd
    monkeypatch.setattr(emu_backend, "inproj_pre_fwd", lambda *a: (calls.append(1), real(*a))[1])
    for on in (True, False):
        monkeypatch.setattr(P, "INPROJ_MFMA", on)
        op.zero_grad(set_to_none=True)
        u = u0.clone().requires_grad_(True)
        y = op(u)
        y.backward(dy)
        res.append([y.float(), u.grad.float()] + [p.grad.float() for _, p in sorted(op.named_parameters())])
    assert len(calls) == 1                          # the kernel ran in the first pass only
    for a, b in zip(*res):
        err = ((a - b).norm() / b.norm().clamp_min(1e-20)).item()
        assert err < 2e-2, err


# ---- the MLP's kernels -------------------------------------------------------------------------------------------------
def _gelu_ref(a):
    return torch.nn.functional.gelu(a.float(), approximate="tanh")


@pytest.mark.parametrize("P,K,N,dtype", [(64, 128, 256, torch.bfloat16), (200, 128, 512, torch.bfloat16), (77, 256, 1024, torch.float16),
                                         (1000, 256, 256, torch.bfloat16), (9, 128, 256, torch.float16),
                                         # runs of several whole tiles per workgroup at both widths (+ a ragged last one): the counted waits of the
                                         # LDS-direct loads are only exercised there -- the emulator retires its memory queue in issue order
                                         (1100, 128, 512, torch.bfloat16), (1088, 128, 256, torch.float16)])
def test_mlp_kernels_vs_autocast_graph(emu_backend, P, K, N, dtype):
    """fc1 + bias + GELU and (dy W2) * GELU'(a) + column sums against the graph they replace, evaluated in fp32 with the roundings
    autocast applies: a = round(x W1^T + b1), h = round(gelu(a)), dh = round(dy W2), da = round(dh * gelu'(a))."""
    _lib = emu_backend
    g = torch.Generator().manual_seed(P + N)
    x = torch.randn(P, K, generator=g).to(dtype)
    W1 = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype)
    b1 = (torch.randn(N, generator=g) * 0.2).to(dtype)
    W2 = (torch.randn(K, N, generator=g) / N ** 0.5).to(dtype)
    dy = torch.randn(P, K, generator=g).to(dtype)
    assert _lib.mlp_supported(P, K, N, dtype)
    a, h = _lib.mlp_fc1_gelu_fwd(x, W1, b1.float())
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    a_ref = x.float() @ W1.float().t() + b1.float()
    assert ((a.float() - a_ref).abs() <= eps * a_ref.abs() + 1e-6).all()
    h_ref = _gelu_ref(a).to(dtype)                                   # from the kernel's own rounded a: the same value or its neighbour
    assert ((h.float() - h_ref.float()).abs() <= 2 * eps * h_ref.float().abs() + 1e-6).all() and (h != h_ref).float().mean() < 0.01
    h64 = torch.nn.functional.gelu(a.double(), approximate="tanh")    # and not further from the exact value than PyTorch's fp32 evaluation
    assert (h.double() - h64).abs().max() <= (h_ref.double() - h64).abs().max() * 1.01 + 1e-9
    da, db1 = _lib.mlp_dh_dgelu_bwd(dy, W2.t().contiguous(), a)
    dh = (dy.float() @ W2.float()).to(dtype)
    af = a.float().requires_grad_(True)
    _gelu_ref(af).backward(dh.float())
    da_ref = af.grad.to(dtype)
    d = (da.float() - da_ref.float()).abs()
    assert (d <= 2 * eps * da_ref.float().abs() + 2 * eps * 1e-2).all()           # (dh itself may differ by an ulp: another summation order)
    assert (da != da_ref).float().mean() < 0.03
    assert ((db1 - da.float().sum(0)).abs() <= 1e-5 * (1 + da.float().abs().sum(0))).all()


def test_fused_mlp_module_matches_the_unfused_graph(emu_backend, monkeypatch):
    """lm.Mlp with and without the kernels (HYENA_FUSED_MLP knob), bf16 tensors: output and every gradient"""
    from functools import partial

    import torch.nn.functional as F

    import hyena_dna_amd.lm as LM
    torch.manual_seed(1)
    mlp = LM.Mlp(128, hidden_features=512, activation=partial(F.gelu, approximate="tanh")).to(torch.bfloat16)
    x0 = torch.randn(3, 50, 128).to(torch.bfloat16)
    dy = torch.randn(3, 50, 128).to(torch.bfloat16)
    calls, real = [], emu_backend.mlp_fc1_gelu_fwd
    monkeypatch.setattr(emu_backend, "mlp_fc1_gelu_fwd", lambda *a: (calls.append(1), real(*a))[1])
    res = []
    for on in (True, False):
        monkeypatch.setattr(LM, "FUSED_MLP", on)
        mlp.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        y = mlp(x)
        y.backward(dy)
        res.append([y.float(), x.grad.float()] + [p.grad.float() for _, p in sorted(mlp.named_parameters())])
    assert len(calls) == 1
    for a, b in zip(*res):
        err = ((a - b).norm() / b.norm().clamp_min(1e-20)).item()
        assert err < 1e-2, err


def test_mlp_with_an_output_width_the_backward_kernel_is_not_built_for_takes_the_library_path(emu_backend):
    """`Mlp(128, hidden_features=256, out_features=64)` passed fc1's check, ran the fused forward and then failed in the fused
    backward (it contracts over fc2.out_features).  The reference's Mlp accepts any out_features: such a module must work end to end."""
    from functools import partial

    import torch.nn.functional as F

    import hyena_dna_amd.lm as LM
    torch.manual_seed(2)
    mlp = LM.Mlp(128, hidden_features=256, out_features=64, activation=partial(F.gelu, approximate="tanh")).to(torch.bfloat16)
    x = torch.randn(2, 40, 128).to(torch.bfloat16).requires_grad_(True)
    assert not mlp._fused_ok(x)
    y = mlp(x)
    y.backward(torch.randn_like(y))
    assert y.shape == (2, 40, 64) and x.grad is not None and mlp.fc2.weight.grad is not None
    assert LM.Mlp(128, hidden_features=512, activation=partial(F.gelu, approximate="tanh")).to(torch.bfloat16)._fused_ok(x)


@pytest.mark.parametrize("P,N,dtype", [(1, 128, torch.bfloat16), (700, 256, torch.bfloat16), (5000, 1024, torch.float16), (33, 64, torch.bfloat16)])
def test_colsum_kernel(emu_backend, P, N, dtype):
    x = torch.randn(P, N, generator=torch.Generator().manual_seed(P)).to(dtype)
    out = emu_backend.colsum(x)
    ref = x.double().sum(0)
    assert out.dtype == torch.float32 and ((out.double() - ref).abs() <= 1e-5 * x.double().abs().sum(0) + 1e-6).all()
    assert torch.equal(out, emu_backend.colsum(x))
    # shapes the kernel does not serve fall back to torch's reduction
    y = torch.randn(10, 96).to(dtype)
    assert torch.allclose(emu_backend.colsum(y), y.float().sum(0))


@pytest.mark.parametrize("B,L,Lx,D,dtype", [(1, 64, 64, 128, torch.bfloat16), (2, 192, 200, 128, torch.float16), (1, 128, 128, 256, torch.bfloat16),
                                            (3, 64, 72, 256, torch.float16), (2, 127, 127, 128, torch.bfloat16), (3, 65, 67, 128, torch.float16),
                                            (1, 255, 255, 256, torch.bfloat16)])
def test_outproj_gate_fwd_vs_cm_post_and_gemm(emu_backend, outproj_gen, B, L, Lx, D, dtype):
    """The fused out_proj kernel (csrc/proj_kernels.h::outproj_gate_fwd_kernel): zT bit-identical to cm_post_fwd, out = the library
    product of that zT with the weight (+ bias), one rounding; with and without the zT side output; L < Lx (l_max cut)."""
    g = torch.Generator().manual_seed(B * 1000 + L + D)
    y = torch.randn(B, D, L, generator=g).to(dtype)
    xT = (torch.randn(3 * D, B, Lx, generator=g) * 0.5).to(dtype)
    bin_ = torch.randn(3 * D, generator=g) * 0.1
    w = torch.randn(3 * D, 3, generator=g) * 0.5
    b = torch.randn(3 * D, generator=g) * 0.1
    W = (torch.randn(D, D, generator=g) / D ** 0.5).to(dtype)
    bias = (torch.randn(D, generator=g) * 0.1).to(dtype).float()
    assert emu_backend.outproj_supported(B, L, Lx, D, dtype) and not emu_backend.outproj_supported(B, 63, 64, D, dtype)
    out, zT = emu_backend.outproj_gate_fwd(y, xT, bin_, w, b, W, bias, want_z=True)
    z_ref = emu_backend.cm_post_fwd(y, xT, bin_, w, b)
    assert torch.equal(zT, z_ref)
    want = (z_ref.reshape(D, B * L).t().float() @ W.float().t() + bias).reshape(B, L, D)
    eps = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    assert out.shape == (B, L, D) and out.dtype == dtype
    assert ((out.float() - want).abs() <= eps * want.abs() + 1e-3 * eps).all()              # one rounding of the fp32 sum
    assert (out != want.to(dtype)).float().mean() < 0.02                                    # (another summation order: a few rounding flips)
    out2, z2 = emu_backend.outproj_gate_fwd(y, xT, None, w, b, W, None, want_z=False)
    assert z2 is None
    want2 = (emu_backend.cm_post_fwd(y, xT, None, w, b).reshape(D, B * L).t().float() @ W.float().t()).reshape(B, L, D)
    assert ((out2.float() - want2).abs() <= eps * want2.abs() + 1e-3 * eps).all()


@pytest.mark.parametrize("B,L,Lx,D,dtype", [(9, 65, 65, 128, torch.bfloat16), (5, 127, 131, 128, torch.float16), (8, 191, 191, 256, torch.bfloat16),
                                            (3, 1023, 1023, 128, torch.bfloat16)])
def test_outproj_rows_at_odd_offsets_in_aligned_pieces(emu_backend, outproj_gen, B, L, Lx, D, dtype):
    """several odd-length sequences per channel row in the flattened layout of _lib.empty_cm (what the operator hands the kernel): rows start at every
    offset (b L + l0) mod 8 elements from a 16-byte boundary -- first / last tile of a sequence, the pulled-back last tile: zT bit-identical to cm_post_fwd,
    out to the product of that zT; and nothing outside the rows is written (the pitch elements behind a channel row keep their NaN fill).
    (Round 6 tried reading / writing such rows in ALIGNED 16-byte pieces shifted in registers: slower, profiles/r6v_outproj_aligned_pieces_not_kept.txt.)"""
    g = torch.Generator().manual_seed(B * 1000 + L + D)
    y = torch.randn(B, D, L, generator=g).to(dtype)
    xT = emu_backend.empty_cm(3 * D, B, Lx, dtype, torch.device("cpu"))
    xT.copy_((torch.randn(3 * D, B, Lx, generator=g) * 0.5).to(dtype))
    assert emu_backend.cm_strides(xT)[0] % 8 == 0 and Lx % 2 == 1
    bin_ = torch.randn(3 * D, generator=g) * 0.1
    w = torch.randn(3 * D, 3, generator=g) * 0.5
    b = torch.randn(3 * D, generator=g) * 0.1
    W = (torch.randn(D, D, generator=g) / D ** 0.5).to(dtype)
    bias = (torch.randn(D, generator=g) * 0.1).to(dtype).float()
    out, zT = emu_backend.outproj_gate_fwd(y, xT, bin_, w, b, W, bias, want_z=True)
    z_ref = emu_backend.cm_post_fwd(y, xT, bin_, w, b)
    assert torch.equal(zT, z_ref)
    want = (z_ref.permute(1, 2, 0).float() @ W.float().t() + bias)
    eps = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    assert ((out.float() - want).abs() <= eps * want.abs() + 1e-3 * eps).all()
    # a zT whose pitch elements are poisoned first: the partial pieces at a tile's two ends must not touch them
    zbuf = emu_backend.empty_cm(D, B, L, dtype, torch.device("cpu"))
    cs = emu_backend.cm_strides(zbuf)[0]
    flat = torch.as_strided(zbuf, (D, cs), (cs, 1))
    flat.fill_(float("nan"))
    import ctypes  # noqa: F401  (the wrapper allocates its own zT; call the C entry point on ours)
    lib = emu_backend.lib()
    out2 = torch.empty(B, L, D, dtype=dtype)
    csx, bsx = emu_backend.cm_strides(xT)
    emu_backend.check(lib.hyena_outproj_gate_fwd_ld(y.data_ptr(), xT.data_ptr(), bin_.data_ptr(), w.data_ptr(), b.data_ptr(), W.data_ptr(), bias.data_ptr(),
                                                    out2.data_ptr(), zbuf.data_ptr(), B, L, Lx, D, csx, bsx, cs, L, L, emu_backend.dtype_code(dtype), None))
    assert torch.equal(zbuf, z_ref) and torch.equal(out2, out)
    assert torch.isnan(flat[:, B * L:]).all()


@pytest.mark.parametrize("L", [128, 136])
def test_operator_with_and_without_the_fused_out_proj(emu_backend, outproj_gen, monkeypatch, L):
    """HyenaOperator (bf16 tensors) through HyenaMixerOutCMFunc vs the round-3 path (cm_post_fwd + library GEMM): output and every
    gradient agree to 16-bit rounding; the kernel really ran; out_proj's weight gradient with and without the saved zT."""
    import hyena_dna_amd.mixer as MX
    from hyena_dna_amd.hyena import HyenaOperator
    torch.manual_seed(4)
    B, D = 2, 128
    op = HyenaOperator(d_model=D, l_max=L, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10)
    with torch.no_grad():
        op.in_proj.bias.normal_(0, 0.1)
        op.out_proj.bias.normal_(0, 0.1)
    op = op.to(torch.bfloat16)
    u0 = torch.randn(B, L, D).to(torch.bfloat16)
    dy = torch.randn(B, L, D).to(torch.bfloat16)
    calls, real = [], emu_backend.outproj_gate_fwd
    monkeypatch.setattr(emu_backend, "outproj_gate_fwd", lambda *a, **k: (calls.append(k.get("want_z")), real(*a, **k))[1])
    res = []
    for on in (True, False):
        monkeypatch.setattr(MX, "OUTPROJ_MFMA", on)
        op.zero_grad(set_to_none=True)
        u = u0.clone().requires_grad_(True)
        y = op(u)
        y.backward(dy)
        res.append([y.float(), u.grad.float()] + [p.grad.float() for _, p in sorted(op.named_parameters())])
    assert calls == [True] and len(res[0]) == len(res[1])          # ran once, keeping zT for out_proj's weight gradient
    for a, b_ in zip(*res):
        err = ((a - b_).norm() / b_.norm().clamp_min(1e-20)).item()
        assert err < 2e-2, err
    # frozen out_proj weight: no zT is written, the other gradients are unchanged
    monkeypatch.setattr(MX, "OUTPROJ_MFMA", True)
    op.out_proj.weight.requires_grad_(False)
    op.zero_grad(set_to_none=True)
    u = u0.clone().requires_grad_(True)
    op(u).backward(dy)
    assert calls == [True, False] and op.out_proj.weight.grad is None
    assert ((u.grad.float() - res[0][1]).norm() / res[0][1].norm()).item() < 1e-6


def test_ragged_length_policy_of_the_fused_out_proj(emu_backend, outproj_gen, monkeypatch):
    """L not a multiple of 8 -- the reference trainer's own lengths (max_length - 1): since round 5 the channel-major tensors are pitched rows, so
    the kernel serves training calls there as well (round 4 routed them to cm_post_fwd + the library GEMM), its choice no longer depends on the
    grad mode, zT's rows start aligned, and outputs and every gradient are the library path's bits."""
    import hyena_dna_amd.mixer as MX
    from hyena_dna_amd.hyena import HyenaOperator
    torch.manual_seed(5)
    B, L, D = 2, 127, 128
    op = HyenaOperator(d_model=D, l_max=L, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10).to(torch.bfloat16)
    u = torch.randn(B, L, D).to(torch.bfloat16)
    calls, zts, real = [], [], emu_backend.outproj_gate_fwd

    def spy(*a, **k):
        calls.append(k.get("want_z"))
        out, zT = real(*a, **k)
        zts.append(zT)
        return out, zT

    monkeypatch.setattr(emu_backend, "outproj_gate_fwd", spy)
    monkeypatch.setattr(MX, "OUTPROJ_MFMA", True)
    res = []
    for fused in (True, False):
        monkeypatch.setattr(MX, "OUTPROJ_MFMA", fused)
        op.zero_grad(set_to_none=True)
        y = op(u)
        y.float().sum().backward()
        res.append([y.detach()] + [p.grad.clone() for p in op.parameters() if p.grad is not None])
    assert calls == [True]                                # training, ragged: the kernel, zT kept for the weight gradient
    assert zts[0].shape == (D, B, L) and zts[0].stride() == (256, L, 1)            # channel rows pitched over the flattened positions (B L = 254 -> 256)
    # y: the kernel's k-ordered fp32 sum against the host library's bf16 GEMM -- the same products, another summation order: identical up to a
    # handful of rounding flips (which host GEMM kernel runs depends on the CPU: this was bit-equal on round 5's build host and is 4 of 32512
    # elements off by one rounding on round 6's); everything downstream of zT -- every gradient -- is the library path's bits
    ya, yb = res[0][0].float(), res[1][0].float()
    assert ((ya - yb).abs() <= 2.0 ** -7 * yb.abs() + 1e-6).all() and (ya != yb).float().mean() < 2e-3
    for a_, b_ in zip(res[0][1:], res[1][1:]):
        assert torch.equal(a_, b_)
    monkeypatch.setattr(MX, "OUTPROJ_MFMA", True)
    with torch.no_grad():
        y_k = op(u)
    assert calls == [True, False]                         # inference: the same kernel, no zT
    assert torch.equal(y_k, res[0][0])


@pytest.mark.parametrize("B,L,D,dtype,with_res", [(2, 127, 128, torch.bfloat16, True), (1, 200, 256, torch.float16, True),
                                                  (3, 64, 128, torch.bfloat16, False), (1, 321, 256, torch.bfloat16, True)])
def test_out_proj_with_the_blocks_add_norm_in_its_epilogue(emu_backend, outproj_gen, monkeypatch, B, L, D, dtype, with_res):
    """Round 5: the prenorm block's second residual add + LayerNorm (simple_lm.py:280-284) inside out_proj's matrix-core kernel
    (hyena_outproj_gate_addnorm_fwd_ld).  The epilogue repeats add_norm_fwd_kernel's arithmetic operation for operation on the rounded
    out_proj output, so hidden, residual' and EVERY gradient are the unfused route's bits (out_proj kernel -> AddLayerNormFunc)."""
    import hyena_dna_amd.hyena as HY
    from hyena_dna_amd.block import dropout_add_layer_norm
    from hyena_dna_amd.hyena import HyenaOperator
    monkeypatch.setattr(HY, "ADD_NORM_FUSED", True)           # (off by default: measured slower on the MI355X, profiles/r5c_outproj_addnorm_not_kept.txt)
    torch.manual_seed(11)
    op = HyenaOperator(d_model=D, l_max=L, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10).to(dtype)
    lw = (1.0 + 0.1 * torch.randn(D)).requires_grad_(True)
    lb = (0.1 * torch.randn(D)).requires_grad_(True)
    u0 = torch.randn(B, L, D).to(dtype)
    r0 = torch.randn(B, L, D) * 3 if with_res else None
    gh, gr = torch.randn(B, L, D).to(dtype), torch.randn(B, L, D)
    res = []
    for fused in (True, False):
        op.zero_grad(set_to_none=True)
        lw.grad = lb.grad = None
        u = u0.clone().requires_grad_(True)
        r = None if r0 is None else r0.clone().requires_grad_(True)
        if fused:
            out = op.forward_add_norm(u, r, lw, lb, 1e-5)
            assert out is not None
            h, rr = out
        else:
            h, rr = dropout_add_layer_norm(op(u), r, lw, lb, 0.0, 1e-5, prenorm=True, residual_in_fp32=True)
        assert h.dtype == dtype and rr.dtype == torch.float32 and h.shape == (B, L, D) and rr.shape == (B, L, D)
        ((h.float() * gh.float()).sum() + (rr * gr).sum()).backward()
        res.append([h.detach(), rr.detach(), u.grad, lw.grad.clone(), lb.grad.clone()] + ([] if r is None else [r.grad]) +
                   [p.grad.clone() for p in op.parameters() if p.grad is not None])
    assert len(res[0]) == len(res[1]) and len(res[0]) > 12
    for a_, b_ in zip(*res):
        assert torch.equal(a_, b_)
    # fp32 operands are not served: the caller falls back to forward + its own add + LayerNorm
    assert op.float().forward_add_norm(u0.float(), r0, lw, lb, 1e-5) is None


def _dgrad_operands(B, L, Lx, D, dtype, seed, exact):
    g = torch.Generator().manual_seed(seed)
    if exact:      # values whose products and 256-term sums are exact in fp32 AND in the 16-bit type: dz^T is then the same in any summation order
        dy2 = torch.randint(-1, 2, (B * L, D), generator=g).to(dtype)
        Wo = torch.randint(-1, 2, (D, D), generator=g).to(dtype)
    else:
        dy2 = torch.randn(B * L, D, generator=g).to(dtype)
        Wo = (torch.randn(D, D, generator=g) / D ** 0.5).to(dtype)
    y = torch.randn(B, D, L, generator=g).to(dtype)
    xT = torch.randn(3 * D, B, Lx, generator=g).to(dtype)
    bin_ = torch.randn(3 * D, generator=g) * 0.1
    w = torch.randn(3 * D, 3, generator=g) * 0.5
    b = torch.randn(3 * D, generator=g) * 0.1
    return dy2, Wo, y, xT, bin_, w, b


@pytest.mark.parametrize("B,L,Lx,D,dtype", [(2, 127, 127, 128, torch.bfloat16), (1, 200, 203, 256, torch.float16), (3, 64, 64, 128, torch.bfloat16),
                                            (1, 2500, 2500, 128, torch.bfloat16), (2, 1100, 1100, 256, torch.bfloat16), (2, 1, 3, 128, torch.float16),
                                            (1, 65, 65, 256, torch.bfloat16)])
def test_out_proj_dgrad_with_the_gate_backward_in_its_epilogue(emu_backend, B, L, Lx, D, dtype):
    """Round 5: dz^T = W_out^T dy^T on the matrix cores with cm_post_bwd's work in the epilogue (hyena_outproj_dgrad_gate_bwd_ld): dz^T is
    never written.  (a) operands whose dz^T is exact in any summation order: d y_conv and d xT are the unfused pair's BITS (library GEMM ->
    cm_post_bwd), the short filter's partial sums agree to summation order; (b) random operands: dz^T differs from the library's by rounding
    flips only -- the outputs agree to a 16-bit ulp almost everywhere.  Runs of several tiles walk DOWN through their sequence with a warm-up
    tile above (L = 2500: three runs), sequences end inside tiles, pieces end inside pieces, rows are pitched."""
    _lib = emu_backend
    from hyena_dna_amd.projection import cm_from_pm
    for exact in (True, False):
        dy2, Wo, y, xT, bin_, w, b = _dgrad_operands(B, L, Lx, D, dtype, seed=L + D + exact, exact=exact)
        yp, xp = _lib.empty_rows((B, D), L, dtype, y.device), _lib.empty_cm(3 * D, B, Lx, dtype, y.device)
        yp.copy_(y)
        xp.copy_(xT)
        assert _lib.outproj_dgrad_supported(B, L, D, dtype)
        dx_f, dx_u = _lib.empty_like_cm(xp).fill_(7.0), _lib.empty_like_cm(xp).fill_(7.0)
        dyc, part0 = _lib.outproj_dgrad_gate_bwd(dy2, Wo.t().contiguous(), yp, xp, bin_, w, b, dx_f)
        dzT = cm_from_pm(Wo.t(), dy2, B, L)
        part = _lib.cm_partials(xp, L)
        dy_u = _lib.cm_post_bwd(dzT, yp, xp, bin_, w, b, dx_u, part)
        assert _lib.ld_of(dyc) == _lib.ld_of(yp)
        assert (dx_f[D:] == 7.0).all() and (dx_f[:D, :, L:] == 7.0).all()              # only rows [0, D), positions < L are written
        red_f, red_u = part0[:, :, :5].sum(1), part[:D, :, :5].sum(1)
        if exact:
            assert torch.equal(dyc, dy_u) and torch.equal(dx_f[:D, :, :L], dx_u[:D, :, :L])
            assert (red_f - red_u).abs().max() <= 2e-5 * red_u.abs().max() + 1e-5
        else:
            for got, ref in ((dyc, dy_u), (dx_f[:D, :, :L], dx_u[:D, :, :L])):
                diff = (got.float() - ref.float()).abs()
                assert (got != ref).float().mean() < 0.03
                assert (diff <= 2.0 ** (-7 if dtype == torch.bfloat16 else -10) * ref.float().abs() + 2e-2).all()
            assert (red_f - red_u).abs().max() <= 2e-2 * red_u.abs().max()


def test_operator_gradients_with_and_without_the_fused_dgrad(emu_backend, monkeypatch):
    """the whole operator, training step: HYENA_OUTPROJ_DGRAD_MFMA on / off give the same gradients (to the rounding flips of dz^T)"""
    import hyena_dna_amd.mixer as MX
    from hyena_dna_amd.hyena import HyenaOperator
    torch.manual_seed(9)
    B, L, D = 2, 191, 128
    op = HyenaOperator(d_model=D, l_max=L, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10).to(torch.bfloat16)
    u0 = torch.randn(B, L, D).to(torch.bfloat16)
    dy = torch.randn(B, L, D).to(torch.bfloat16)
    res = []
    # "auto" (round 6, profiles/r6o_bench_dgrad.txt): two or more sequences, except the very short rows of the d_model 128 models; never at B = 1
    assert MX._dgrad_fused(2, 4096, 256, torch.bfloat16) and MX._dgrad_fused(8, 32767, 256, torch.bfloat16) and MX._dgrad_fused(4, 4096, 128, torch.bfloat16)
    assert not MX._dgrad_fused(1, 1 << 20, 256, torch.bfloat16) and not MX._dgrad_fused(8, 191, 128, torch.bfloat16)
    assert not MX._dgrad_fused(8, 191, 128, torch.float32)
    for on in (True, False):
        monkeypatch.setattr(MX, "DGRAD_MFMA", on)
        op.zero_grad(set_to_none=True)
        u = u0.clone().requires_grad_(True)
        op(u).backward(dy)
        res.append({"du": u.grad.float(), **{n: p.grad.float() for n, p in op.named_parameters() if p.grad is not None}})
    assert res[0].keys() == res[1].keys() and len(res[0]) > 10
    for n in res[0]:
        a_, b_ = res[0][n], res[1][n]
        assert ((a_ - b_).norm() / b_.norm().clamp_min(1e-20)).item() < 2e-2, n


def test_several_odd_length_sequences_are_padded_inside_in_proj(emu_backend, monkeypatch):
    """B > 1 sequences of a length that is not a multiple of 64 (the reference trainer's max_length - 1): projection.InProjPreCMFunc runs the
    kernels on zero-padded sequences, so every channel-major row starts aligned.  Same values: the output and the input gradient bit for bit,
    the parameter gradients to the summation order of the position sums (the padded positions add exact zeros)."""
    import hyena_dna_amd.projection as P
    from hyena_dna_amd.hyena import HyenaOperator
    torch.manual_seed(21)
    B, L, D = 3, 321, 128
    op = HyenaOperator(d_model=D, l_max=L + 2, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10).to(torch.bfloat16)
    u0 = torch.randn(B, L, D).to(torch.bfloat16)
    dy = torch.randn(B, L, D).to(torch.bfloat16)
    seen, real = [], emu_backend.inproj_pre_fwd
    monkeypatch.setattr(emu_backend, "inproj_pre_fwd", lambda u, *a: (seen.append(tuple(u.shape)), real(u, *a))[1])
    res = []
    for pad in (True, False):
        monkeypatch.setattr(P, "PAD_SEQUENCES", pad)
        op.zero_grad(set_to_none=True)
        u = u0.clone().requires_grad_(True)
        y = op(u)
        y.backward(dy)
        res.append((y.detach(), u.grad, {n: p.grad.float() for n, p in op.named_parameters() if p.grad is not None}))
    assert seen == [(B, 384, D), (B, L, D)]
    assert res[0][0].shape == (B, L, D) and torch.equal(res[0][0], res[1][0]) and res[0][1].shape == (B, L, D) and torch.equal(res[0][1], res[1][1])
    assert res[0][2].keys() == res[1][2].keys() and len(res[0][2]) > 10
    for n in res[0][2]:
        a_, b_ = res[0][2][n], res[1][2][n]
        assert ((a_ - b_).norm() / b_.norm().clamp_min(1e-20)).item() < 1e-2, n           # (16-bit parameter gradients: one rounding of slightly different fp32 sums)
