"""The matrix-core input projection with the front of the shell in its epilogue (csrc/proj_kernels.h) on a real MI355X, through
the C ABI: xT against the library GEMM on the same 16-bit operands (one rounding of an fp32 sum either way) and against the fp64
product, vg BIT-IDENTICAL to cm_pre_fwd on the kernel's own xT, at the contract shapes incl. an odd length (rows start
under-aligned) and a truncated one (Lc < Lx), determinism."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[1, 2])
def outproj_gen(request, gpu_lib):
    """both generations of the out_proj forward kernel (include/hyena_proj.h, hyena_proj_kernel_generation): round 4's and round 6's (default)"""
    prev = gpu_lib.proj_kernel_generation(0)
    assert gpu_lib.proj_kernel_generation(0, request.param) == request.param
    yield request.param
    torch.cuda.synchronize()
    gpu_lib.proj_kernel_generation(0, prev)


@pytest.fixture(params=[1, 2])
def inproj_gen(request, gpu_lib):
    """both generations of the in_proj forward kernel: rounds 3 / 4's and round 6's (default)"""
    prev = gpu_lib.proj_kernel_generation(1)
    assert gpu_lib.proj_kernel_generation(1, request.param) == request.param
    yield request.param
    torch.cuda.synchronize()
    gpu_lib.proj_kernel_generation(1, prev)


@pytest.mark.parametrize("B,Lx,Lc,D,dtype", [(8, 1024, 1024, 128, torch.bfloat16), (8, 32768, 32768, 256, torch.bfloat16),
                                             (2, 160000, 160000, 256, torch.bfloat16), (1, 1048576, 1048576, 256, torch.bfloat16),
                                             (1, 999999, 999999, 256, torch.bfloat16), (3, 4099, 4000, 128, torch.float16),
                                             (2, 70001, 70001, 256, torch.float16), (5, 9, 9, 128, torch.bfloat16)])
def test_inproj_pre_fwd_on_gpu(gpu_lib, inproj_gen, B, Lx, Lc, D, dtype):
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(Lx + D)
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)      # noqa: E731
    u = rn(B, Lx, D).to(dtype)
    W = (rn(3 * D, D) / D ** 0.5).to(dtype)
    bin_, w, b = rn(3 * D) * 0.3, rn(3 * D, 3) * 0.5, rn(3 * D) * 0.2
    assert gpu_lib.proj_supported(B, Lx, D, dtype)
    xT, vg = gpu_lib.inproj_pre_fwd(u, W, bin_, w, b, Lc)
    lib = torch.mm(W, u.reshape(B * Lx, D).t()).view(3 * D, B, Lx)              # hipBLASLt, fp32 accumulation, one rounding
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    diff = (xT.float() - lib.float()).abs()
    assert (diff <= 2 * eps * lib.float().abs() + 1e-6).all()                   # at most one ulp apart ...
    assert (xT != lib).float().mean().item() < 0.02                              # ... and almost everywhere identical
    # a slice against the fp64 product
    rows = torch.arange(0, 3 * D, 37, device=dev)
    ref = torch.mm(W[rows].double(), u.reshape(B * Lx, D)[: 4096].double().t())
    got = xT.reshape(3 * D, B * Lx)[rows][:, : 4096].double()
    assert ((got - ref).abs() <= eps * ref.abs() + 1e-6).all()
    # the same expression, FMA by FMA, as cm_pre_fwd on the kernel's own xT: identical values -- on the gfx950 binary up to ~1e-6 of
    # the fp16 results landing on the neighbouring value (one fp32 ulp somewhere in the two kernels' compiled arithmetic; under
    # tests/hipemu the two are bit-identical, tests/test_proj_emu.py)
    vg_ref = gpu_lib.cm_pre_fwd(xT, bin_, w, b, Lc)
    neq = vg != vg_ref
    assert neq.float().mean().item() <= 2e-5
    assert ((vg.float() - vg_ref.float()).abs() <= 2 * eps * vg_ref.float().abs() + 1e-7).all()
    xT2, vg2 = gpu_lib.inproj_pre_fwd(u, W, bin_, w, b, Lc)
    assert torch.equal(xT, xT2) and torch.equal(vg, vg2)


def test_operator_uses_the_mfma_projection_and_matches_the_library_path(gpu_lib, monkeypatch):
    import hyena_dna_amd.projection as P
    from hyena_dna_amd.hyena import HyenaOperator
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    B, L, D = 2, 40000, 256
    op = HyenaOperator(d_model=D, l_max=L + 2, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10,
                       lr=6e-4, wd=0.0, lr_pos_emb=0.0).to(dev)
    u0 = torch.randn(B, L, D, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(B, L, D, device=dev, dtype=torch.bfloat16)
    calls, real = [], gpu_lib.inproj_pre_fwd
    monkeypatch.setattr(gpu_lib, "inproj_pre_fwd", lambda *a: (calls.append(1), real(*a))[1])
    res = []
    for on in (True, False):
        monkeypatch.setattr(P, "INPROJ_MFMA", on)
        op.zero_grad(set_to_none=True)
        u = u0.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = op(u)
        y.backward(dy)
        res.append([y.float(), u.grad.float()] + [p.grad.float() for _, p in sorted(op.named_parameters())])
    assert len(calls) == 1
    for a, b in zip(*res):
        err = ((a - b).norm() / b.norm().clamp_min(1e-20)).item()
        assert err < 1e-2, err


@pytest.mark.parametrize("P,K,N,dtype", [(8 * 1024, 128, 512, torch.bfloat16), (8 * 32768, 256, 1024, torch.bfloat16), (1048576, 256, 1024, torch.bfloat16),
                                         (999999, 256, 1024, torch.float16), (77, 128, 256, torch.bfloat16)])
def test_mlp_kernels_on_gpu(gpu_lib, P, K, N, dtype):
    """fc1 + bias + GELU and (dy W2) * GELU'(a) + column sums (csrc/proj_kernels.h mlp_kernel) against the autocast graph they
    replace: library GEMMs + PyTorch's tanh-GELU forward / backward on the same 16-bit tensors."""
    F = torch.nn.functional
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(P % 1000 + N)
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)      # noqa: E731
    x = rn(P, K).to(dtype)
    W1, b1 = (rn(N, K) / K ** 0.5).to(dtype), (rn(N) * 0.2).to(dtype)
    W2 = (rn(K, N) / N ** 0.5).to(dtype)
    dy = rn(P, K).to(dtype)
    assert gpu_lib.mlp_supported(P, K, N, dtype)
    a, h = gpu_lib.mlp_fc1_gelu_fwd(x, W1, b1.float())
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    a_lib = F.linear(x, W1, b1)
    assert ((a.float() - a_lib.float()).abs() <= 2 * eps * a_lib.float().abs() + 1e-5).all() and (a != a_lib).float().mean().item() < 0.02
    h_ref = F.gelu(a, approximate="tanh")                       # PyTorch's kernel on the kernel's own a
    assert ((h.float() - h_ref.float()).abs() <= 2 * eps * h_ref.float().abs() + 1e-6).all() and (h != h_ref).float().mean().item() < 0.01
    da, db1 = gpu_lib.mlp_dh_dgelu_bwd(dy, W2.t().contiguous(), a)
    dh = torch.mm(dy, W2)
    a_ = a.clone().requires_grad_(True)
    F.gelu(a_, approximate="tanh").backward(dh)
    da_ref = a_.grad
    err = ((da.float() - da_ref.float()).norm() / da_ref.float().norm()).item()
    assert err < 3e-3, err                                       # 16-bit roundings of dh on either side
    assert ((da.float() - da_ref.float()).abs() <= 4 * eps * da_ref.float().abs() + 4 * eps * 0.05).all()
    ref_db1 = da.float().sum(0)
    assert ((db1 - ref_db1).abs() <= 1e-4 * (1 + da.float().abs().sum(0))).all()
    a2, h2 = gpu_lib.mlp_fc1_gelu_fwd(x, W1, b1.float())
    da2, db2 = gpu_lib.mlp_dh_dgelu_bwd(dy, W2.t().contiguous(), a)
    assert torch.equal(a, a2) and torch.equal(h, h2) and torch.equal(da, da2) and torch.equal(db1, db2)


def test_lm_step_with_and_without_the_fused_mlp(gpu_lib, monkeypatch):
    """HyenaDNALM loss and gradients, bf16 autocast, with the MFMA MLP kernels vs two library GEMMs + PyTorch's GELU"""
    import hyena_dna_amd.lm as LM
    dev = torch.device("cuda", 0)
    L, B, D = 8192, 2, 128
    layer = dict(l_max=L + 2, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10)
    torch.manual_seed(3)
    m = LM.HyenaDNALM(d_model=D, n_layer=2, d_inner=4 * D, vocab_size=12, layer=layer, resid_dropout=0.0, embed_dropout=0.0,
                      pad_vocab_size_multiple=8).to(dev)
    ids = torch.randint(7, 11, (B, L), device=dev)
    res = []
    for on in (True, False):
        monkeypatch.setattr(LM, "FUSED_MLP", on)
        m.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = m.loss(ids, torch.roll(ids, -1, 1))
        loss.backward()
        res.append((loss.item(), {n: p.grad.float().clone() for n, p in m.named_parameters()}))
    assert abs(res[0][0] - res[1][0]) < 2e-3 * abs(res[1][0])
    for n, gref in res[1][1].items():
        e = ((res[0][1][n] - gref).norm() / gref.norm().clamp_min(1e-20)).item()
        assert e < 3e-2, (n, e)


@pytest.mark.parametrize("P,N,dtype", [(1048576, 256, torch.bfloat16), (262144, 1024, torch.float16), (999999, 128, torch.bfloat16), (7, 256, torch.bfloat16)])
def test_colsum_on_gpu(gpu_lib, P, N, dtype):
    """the bias-gradient column sums (csrc/proj_kernels.h colsum_kernel) against the fp64 sums; deterministic"""
    dev = torch.device("cuda", 0)
    x = torch.randn(P, N, device=dev, generator=torch.Generator(device=dev).manual_seed(P % 997)).to(dtype)
    out = gpu_lib.colsum(x)
    ref = x.double().sum(0)
    assert out.dtype == torch.float32 and ((out.double() - ref).abs() <= 2e-6 * x.double().abs().sum(0) + 1e-5).all()
    assert torch.equal(out, gpu_lib.colsum(x))


@pytest.mark.parametrize("B,L,Lx,D,dtype", [(8, 1024, 1024, 128, torch.bfloat16), (8, 32768, 32768, 256, torch.bfloat16),
                                            (2, 160000, 160000, 256, torch.bfloat16), (1, 1048576, 1048576, 256, torch.bfloat16),
                                            (3, 4096, 4104, 128, torch.float16), (2, 70016, 70016, 256, torch.float16),
                                            (1, 999999, 999999, 256, torch.bfloat16), (3, 4099, 4101, 128, torch.float16), (2, 32767, 32767, 256, torch.bfloat16)])
def test_outproj_gate_fwd_on_gpu(gpu_lib, outproj_gen, B, L, Lx, D, dtype):
    """The fused out_proj kernel (round 4) on the MI355X through the C ABI: zT BIT-IDENTICAL to cm_post_fwd, out against the library
    GEMM on that zT (one rounding of an fp32 sum either way: at most an ulp apart, almost everywhere identical), a slice against the
    fp64 product, determinism, the contract shapes."""
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(L + D)
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)      # noqa: E731
    y = rn(B, D, L).to(dtype)
    xT = (rn(3 * D, B, Lx) * 0.5).to(dtype)
    bin_, w, b = rn(3 * D) * 0.1, rn(3 * D, 3) * 0.5, rn(3 * D) * 0.1
    W = (rn(D, D) / D ** 0.5).to(dtype)
    bias = (rn(D) * 0.1).to(dtype)
    assert gpu_lib.outproj_supported(B, L, Lx, D, dtype)
    out, zT = gpu_lib.outproj_gate_fwd(y, xT, bin_, w, b, W, bias.float(), want_z=True)
    z_ref = gpu_lib.cm_post_fwd(y, xT, bin_, w, b)
    assert torch.equal(zT, z_ref)
    lib = torch.addmm(bias, z_ref.reshape(D, B * L).t(), W.t()).view(B, L, D)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert ((out.float() - lib.float()).abs() <= 2 * eps * lib.float().abs() + 1e-5).all()
    assert (out != lib).float().mean().item() < 0.02
    pos = torch.arange(0, B * L, max(1, B * L // 997), device=dev)
    z64 = z_ref.reshape(D, B * L)[:, pos].t().double()
    ref = z64 @ W.double().t() + bias.double()
    got = out.reshape(B * L, D)[pos].double()
    assert ((got - ref).abs() <= 1.01 * eps * ref.abs() + 2e-6 * z64.abs().sum(1, keepdim=True) + 1e-30).all()
    out2, z2 = gpu_lib.outproj_gate_fwd(y, xT, bin_, w, b, W, bias.float(), want_z=False)
    assert z2 is None and torch.equal(out2, out)


@pytest.mark.parametrize("L", [8192, 8200])
def test_operator_uses_the_fused_out_proj_and_matches_the_library_path(gpu_lib, monkeypatch, L):
    """HyenaOperator under bf16 autocast: HyenaMixerOutCMFunc (the kernel really runs) vs HYENA_OUTPROJ_MFMA=0, output and every
    gradient to 16-bit rounding."""
    import hyena_dna_amd.mixer as MX
    from hyena_dna_amd.hyena import HyenaOperator
    dev = torch.device("cuda", 0)
    torch.manual_seed(6)
    B, D = 2, 256
    op = HyenaOperator(d_model=D, l_max=L, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10).to(dev)
    u0 = torch.randn(B, L, D, device=dev)
    dy = torch.randn(B, L, D, device=dev)
    calls, real = [], gpu_lib.outproj_gate_fwd
    monkeypatch.setattr(gpu_lib, "outproj_gate_fwd", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    res = []
    for on in (True, False):
        monkeypatch.setattr(MX, "OUTPROJ_MFMA", on)
        op.zero_grad(set_to_none=True)
        u = u0.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            yy = op(u)
        yy.float().backward(dy)
        res.append([yy.float(), u.grad.float()] + [p.grad.float() for _, p in sorted(op.named_parameters()) if p.grad is not None])
    assert len(calls) == 1 and len(res[0]) == len(res[1])
    for a, b_ in zip(*res):
        assert ((a - b_).norm() / b_.norm().clamp_min(1e-20)).item() < 1.5e-2


@pytest.mark.parametrize("B,L,D,dtype", [(8, 32767, 256, torch.bfloat16), (2, 159999, 256, torch.bfloat16), (1, 1048575, 256, torch.bfloat16),
                                         (1, 1048576, 256, torch.bfloat16), (3, 4099, 128, torch.float16), (2, 65, 128, torch.bfloat16)])
def test_out_proj_dgrad_with_the_gate_backward_on_gpu(gpu_lib, B, L, D, dtype):
    """hyena_outproj_dgrad_gate_bwd_ld (round 5) against the pair it replaces -- library GEMM dz^T = W_out^T dy^T, then cm_post_bwd -- on the
    gfx950 binary at the contract shapes incl. the reference trainer's odd lengths: with operands whose dz^T is exact in any summation
    order d y_conv and d xT are the pair's bits; with random operands they agree to the rounding flips of dz^T; determinism."""
    from hyena_dna_amd.projection import cm_from_pm
    dev = torch.device("cuda", 0)
    for exact in (True, False):
        g = torch.Generator(device=dev).manual_seed(L + D + exact)
        rn = lambda *s: torch.randn(*s, generator=g, device=dev)      # noqa: E731
        if exact:
            dy2 = torch.randint(-1, 2, (B * L, D), generator=g, device=dev).to(dtype)
            Wo = torch.randint(-1, 2, (D, D), generator=g, device=dev).to(dtype)
        else:
            dy2, Wo = rn(B * L, D).to(dtype), (rn(D, D) / D ** 0.5).to(dtype)
        y = gpu_lib.empty_rows((B, D), L, dtype, dev).copy_(rn(B, D, L).to(dtype))
        xT = gpu_lib.empty_cm(3 * D, B, L, dtype, dev).copy_(rn(3 * D, B, L).to(dtype))
        bin_, w, b = rn(3 * D) * 0.1, rn(3 * D, 3) * 0.5, rn(3 * D) * 0.1
        dx_f, dx_u = gpu_lib.empty_like_cm(xT).fill_(7.0), gpu_lib.empty_like_cm(xT).fill_(7.0)
        dyc, part0 = gpu_lib.outproj_dgrad_gate_bwd(dy2, Wo.t().contiguous(), y, xT, bin_, w, b, dx_f)
        dzT = cm_from_pm(Wo.t(), dy2, B, L)
        part = gpu_lib.cm_partials(xT, L)
        dy_u = gpu_lib.cm_post_bwd(dzT, y, xT, bin_, w, b, dx_u, part)
        assert (dx_f[D:] == 7.0).all()
        red_f, red_u = part0[:, :, :5].sum(1), part[:D, :, :5].sum(1)
        if exact:
            if dtype == torch.float16:
                # (fp16 stores: hipcc folds a multiply and the conversion into ONE rounding -- v_fma_mixlo_f16 -- in one kernel and not in the
                #  other: the neighbouring fp16 value in a few elements per million, as in tests/test_gpu_block.py; bit-identical under tests/hipemu)
                for got, ref in ((dyc, dy_u), (dx_f[:D], dx_u[:D])):
                    assert (got != ref).float().mean().item() < 1e-3
                    assert ((got.float() - ref.float()).abs() <= 2.0 ** -10 * ref.float().abs() + 1e-6).all()
            else:
                assert torch.equal(dyc, dy_u) and torch.equal(dx_f[:D], dx_u[:D])
            assert (red_f - red_u).abs().max() <= 1e-4 * red_u.abs().max() + 1e-3
            dx_2 = gpu_lib.empty_like_cm(xT).fill_(7.0)
            dyc2, part2 = gpu_lib.outproj_dgrad_gate_bwd(dy2, Wo.t().contiguous(), y, xT, bin_, w, b, dx_2)
            assert torch.equal(dyc, dyc2) and torch.equal(dx_f[:D], dx_2[:D]) and torch.equal(part0[..., :5], part2[..., :5])      # (floats 5 - 7 of a record are padding, never written)
        else:
            eps = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
            for got, ref in ((dyc, dy_u), (dx_f[:D], dx_u[:D])):
                assert (got != ref).float().mean().item() < 0.03
                assert ((got.float() - ref.float()).abs() <= eps * ref.float().abs() + 2e-2).all()
            assert (red_f - red_u).abs().max() <= 2e-2 * red_u.abs().max()


@pytest.mark.parametrize("B,L,D,dtype", [(8, 32767, 256, torch.bfloat16), (1, 1048575, 256, torch.bfloat16), (2, 4099, 128, torch.float16)])
def test_out_proj_with_add_norm_epilogue_on_gpu(gpu_lib, outproj_gen, B, L, D, dtype):
    """hyena_outproj_gate_addnorm_fwd_ld (round 5; off by default -- measured slower, profiles/r5c_outproj_addnorm_not_kept.txt): all five outputs
    are the bits of hyena_outproj_gate_fwd_ld followed by hyena_add_norm_fwd"""
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(L + D)
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)      # noqa: E731
    y = gpu_lib.empty_rows((B, D), L, dtype, dev).copy_(rn(B, D, L).to(dtype))
    xT = gpu_lib.empty_cm(3 * D, B, L, dtype, dev).copy_(rn(3 * D, B, L).to(dtype))
    bin_, w, b = rn(3 * D) * 0.1, rn(3 * D, 3) * 0.5, rn(3 * D) * 0.1
    W = (rn(D, D) / D ** 0.5).to(dtype)
    bias = (rn(D) * 0.1).to(dtype).float()
    res, lw, lb = rn(B * L, D) * 3, 1.0 + 0.1 * rn(D), 0.1 * rn(D)
    o, z = gpu_lib.outproj_gate_fwd(y, xT, bin_, w, b, W, bias, want_z=True)
    two = gpu_lib.add_norm_fwd(o.view(B * L, D), res, lw, lb, 1e-5, dtype) + (z,)
    one = gpu_lib.outproj_gate_addnorm_fwd(y, xT, bin_, w, b, W, bias, True, res, lw, lb, 1e-5)
    for p, q in zip(two, one):
        assert torch.equal(p.reshape(-1), q.reshape(-1))


@pytest.mark.parametrize("B,L", [(1, 1048575), (8, 32767), (2, 159999), (1, 70001), (3, 4099)])
def test_weight_gradient_slicing_at_odd_position_counts(gpu_lib, B, L):
    """projection.split_plan / _tail_product (round 5): the three weight-gradient products at the reference trainer's odd position counts --
    first-level slices of a multiple of 64 rows, the remainder in 256-row slices, the last < 256 rows as a masked 16-bit slice -- against
    the fp64 products, and twice for determinism."""
    from hyena_dna_amd.projection import split_k_weight_grad, split_plan, wgrad_cm_pm, wgrad_pm_cm
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(B * L)
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)      # noqa: E731
    P, D = B * L, 256
    levels, done = split_plan(P, 3 * D * D)
    assert 0 < P - done < 256 or P % 256 == 0
    dy2, x2 = rn(P, D).bfloat16(), rn(P, D).bfloat16()
    dT = gpu_lib.empty_cm(3 * D, B, L, torch.bfloat16, dev).copy_(rn(3 * D, B, L).bfloat16())
    zT = gpu_lib.empty_cm(D, B, L, torch.bfloat16, dev).copy_(rn(D, B, L).bfloat16())
    got = [split_k_weight_grad(dy2, x2), wgrad_cm_pm(dT, x2), wgrad_pm_cm(dy2, zT)]
    again = [split_k_weight_grad(dy2, x2), wgrad_cm_pm(dT, x2), wgrad_pm_cm(dy2, zT)]
    d2 = dT.permute(0, 1, 2).reshape(3 * D, P).double()
    z2 = zT.reshape(D, P).double()
    ref = [dy2.double().t() @ x2.double(), d2 @ x2.double(), dy2.double().t() @ z2.t()]
    for a, b, r in zip(got, again, ref):
        assert a.dtype == torch.float32 and a.shape == r.shape and torch.equal(a, b)
        assert ((a.double() - r).norm() / r.norm()).item() < 2e-6
