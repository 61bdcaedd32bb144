"""Parity on a real MI355X: the gfx950 library, called through the C ABI (hyena_dna_amd._lib -> ctypes), against the
oracle (CPU restatement of the reference, pinned to reference-minted golden vectors), plus size-independent
properties at the BASELINE sizes where the CPU oracle would take too long."""
import pytest
import torch

from oracle import hyena_oracle as O

pytestmark = pytest.mark.gpu

REL_FP32 = 3e-6        # rel-L2; north_star tolerance is 1e-3, fp32 mode is expected <= 1e-5


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _inputs(B, D, L, dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    u = torch.randn(B, D, L, generator=g).to(dtype)
    k = torch.randn(D, L, generator=g) * torch.exp(-5.0 * torch.linspace(0, 1, L))[None] * 0.1
    bias = torch.randn(D, generator=g)
    dout = torch.randn(B, D, L, generator=g).to(dtype)
    return u, k, bias, dout


def _oracle(u, k, bias, dout):
    u_ = u.clone().requires_grad_(True)
    k_ = k.clone().requires_grad_(True)
    b_ = bias.clone().requires_grad_(True)
    out = O.fftconv_ref(u_, k_, b_)
    out.backward(dout)
    return out.detach(), u_.grad, k_.grad, b_.grad


def _gpu(gpu_lib, u, k, bias, dout, chunk=None):
    dev = torch.device("cuda", 0)
    ud, kd, bd, gd = u.to(dev), k.to(dev), bias.to(dev), dout.to(dev)
    out = gpu_lib.fftconv_fwd(ud, kd, bd, chunk=chunk)
    du, dk, dbias = gpu_lib.fftconv_bwd(gd, ud, kd, bd, chunk=chunk)
    torch.cuda.synchronize()
    return out.cpu(), du.cpu(), dk.cpu(), dbias.cpu()


def test_native_library_is_the_one_loaded(gpu_lib):
    import os
    assert os.path.basename(gpu_lib.LIB_PATH) == "libhyena_fftconv.so"
    maps = open("/proc/self/maps").read()
    assert "libhyena_fftconv.so" in maps and "_emu_" not in maps


@pytest.mark.parametrize("B,D,L,chunk", [
    (2, 3, 1, None), (2, 3, 8, None), (3, 5, 1023, None), (8, 128, 1024, None), (2, 4, 1025, 3), (2, 3, 2048, None),
    (1, 3, 5000, None), (2, 2, 8191, None), (1, 4, 16384, 1), (2, 8, 32768, None), (1, 2, 65536, None),
    (1, 3, 160000, 2), (1, 2, 450560, None), (1, 2, 1048576, None), (1, 1, 1048575, None),
    # column sizes that are not powers of two (M1 = 3, 5, 7, 12, 28, 96, 224, 384, 640, 768)
    (2, 3, 3069, None), (2, 2, 5117, None), (2, 2, 7165, 1), (1, 3, 12285, None), (1, 2, 28669, None), (1, 2, 98301, None),
    (1, 2, 229373, None), (1, 1, 393213, None), (1, 1, 655357, None), (1, 1, 786429, None),
])
def test_fp32_fwd_bwd_vs_oracle(gpu_lib, B, D, L, chunk):
    u, k, bias, dout = _inputs(B, D, L, torch.float32, seed=L)
    out, du, dk, dbias = _gpu(gpu_lib, u, k, bias, dout, chunk)
    r_out, r_du, r_dk, r_db = _oracle(u, k, bias, dout)
    assert _rel(out, r_out) < REL_FP32
    assert _rel(du, r_du) < REL_FP32
    assert _rel(dk, r_dk) < REL_FP32
    assert _rel(dbias, r_db) < 1e-5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,D,L", [(2, 3, 37), (2, 4, 1023), (2, 8, 32768), (1, 2, 160000)])
def test_half_io_vs_oracle(gpu_lib, dtype, B, D, L):
    u, k, bias, dout = _inputs(B, D, L, dtype, seed=L + 1)
    out, du, dk, dbias = _gpu(gpu_lib, u, k, bias, dout)
    r_out, r_du, r_dk, r_db = _oracle(u, k, bias, dout)
    assert out.dtype == dtype and du.dtype == dtype and dk.dtype == torch.float32
    eps = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    diff = (out.float() - r_out.float()).abs()
    assert (diff <= eps * r_out.float().abs() + 2e-5).all()          # at most one 16-bit ulp from the reference
    assert (out != r_out).float().mean() < 0.02
    t_out, t_du, _, _ = _oracle(u.float(), k, bias, dout.float())      # fp32 result on the same 16-bit inputs
    assert ((out.float() - t_out).abs() <= 0.5 * eps * t_out.abs() * 1.001 + 2e-5).all()
    assert ((du.float() - t_du).abs() <= 0.5 * eps * t_du.abs() * 1.001 + 2e-5).all()
    assert _rel(du.float(), r_du.float()) < 1.5 * eps
    assert _rel(dk, r_dk) < REL_FP32 and _rel(dbias, r_db) < 1e-5
    assert _rel(out.float(), r_out.float()) < 1e-3                    # the north_star tolerance


@pytest.mark.parametrize("name", ["b2d4l8", "b2d3l37", "b1d4l1023", "b2d4l1024", "b2d4l1024_5d", "b2d4l1000_bf16",
                                  "b1d2l4100"])
def test_golden_vectors_from_reference(gpu_lib, golden_fftconv, name):
    c = golden_fftconv[name]
    out, du, dk, dbias = _gpu(gpu_lib, c["u"], c["k"], c["bias"], c["dout"])
    if c["u"].dtype == torch.float32:
        assert _rel(out, c["out"]) < REL_FP32 and _rel(du, c["du"]) < REL_FP32
    else:
        assert _rel(out.float(), c["out"].float()) < 1e-3 and _rel(du.float(), c["du"].float()) < 1.2e-2
    assert _rel(dk, c["dk"]) < REL_FP32 and _rel(dbias, c["dbias"]) < 1e-5


@pytest.mark.parametrize("name", ["b1d2l40000", "b1d1l160000_bf16"])
def test_golden_vectors_from_reference_large(gpu_lib, golden_fftconv_large, name):
    """reference-minted vectors at a two-stage column size (L = 40000, M1 = 64) and a mixed-radix one (L = 160000, M1 = 160)"""
    c = golden_fftconv_large[name]
    out, du, dk, dbias = _gpu(gpu_lib, c["u"], c["k"], c["bias"], c["dout"])
    if c["u"].dtype == torch.float32:
        assert _rel(out, c["out"]) < REL_FP32 and _rel(du, c["du"]) < REL_FP32
    else:
        eps = 2.0 ** -7
        diff = (out.float() - c["out"].float()).abs()
        assert (diff <= eps * c["out"].float().abs() + 2e-5).all() and (out != c["out"]).float().mean() < 0.02
        assert _rel(du.float(), c["du"].float()) < 1.5 * eps
    assert _rel(dk, c["dk"]) < REL_FP32 and _rel(dbias, c["dbias"]) < 2e-5


def test_autograd_function_5d_on_gpu(gpu_lib):
    from hyena_dna_amd.fftconv import fftconv_func
    dev = torch.device("cuda", 0)
    B, D, L = 2, 16, 3000
    u, k, bias, dout = _inputs(B, D, L, torch.float32, seed=11)
    u_ = u.to(dev).requires_grad_(True)
    k_ = k.to(dev).requires_grad_(True)
    b_ = bias.to(dev).requires_grad_(True)
    out = fftconv_func(u_.reshape(B, 1, D, 1, L), k_, b_[None, :, None], gelu=False)
    out.backward(dout.to(dev).reshape(B, 1, D, 1, L))
    r_out, r_du, r_dk, r_db = _oracle(u, k, bias, dout)
    assert _rel(out.reshape(B, D, L), r_out) < REL_FP32 and _rel(u_.grad, r_du) < REL_FP32
    assert _rel(k_.grad, r_dk) < REL_FP32 and _rel(b_.grad, r_db) < 1e-5


@pytest.mark.parametrize("name", ["d8l64", "d16l257", "d8l80_trunc"])
def test_operator_mirror_on_gpu(gpu_lib, golden_operator, name):
    from hyena_dna_amd.hyena import HyenaOperator
    dev = torch.device("cuda", 0)
    c = golden_operator[name]
    op = HyenaOperator(d_model=c["d_model"], l_max=c["l_max"], order=2, filter_order=64, emb_dim=5,
                       short_filter_order=3, modulate=True, w=10, lr=6e-4, wd=0.0, lr_pos_emb=0.0)
    op.load_state_dict(c["state_dict"])
    op = op.to(dev)
    u = c["u"].to(dev).requires_grad_(True)
    y = op(u)
    y.backward(c["dy"].to(dev))
    torch.testing.assert_close(y.cpu(), c["y"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(u.grad.cpu(), c["du"], rtol=1e-3, atol=1e-5)
    for n, p in op.named_parameters():
        torch.testing.assert_close(p.grad.cpu(), c["grads"][n], rtol=2e-3, atol=1e-4, msg=lambda m, n=n: f"{n}: {m}")


# ---- properties at the BASELINE sizes (no CPU oracle needed) -------------------------------------------------
@pytest.mark.parametrize("L,D,B,dtype", [(32768, 256, 8, torch.bfloat16), (160000, 256, 2, torch.bfloat16),
                                          (450560, 256, 1, torch.bfloat16), (1048576, 256, 1, torch.bfloat16),
                                          (1048576, 32, 1, torch.float32)])
def test_properties_at_baseline_sizes(gpu_lib, L, D, B, dtype):
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(L + D)          # device RNG: host randn of 2.7e8 values is slow
    rn = lambda *shape: torch.randn(*shape, generator=g, device=dev)   # noqa: E731
    k = rn(D, L) * torch.exp(-5.0 * torch.linspace(0, 1, L, device=dev))[None] * 0.1
    bias = rn(D)
    # (1) impulse response: u = delta at position p  ->  out[t] = k[t - p] + bias * delta  (exact up to rounding)
    p = 12345 % L
    u = torch.zeros(B, D, L, dtype=dtype, device=dev)
    u[:, :, p] = 1.0
    out = gpu_lib.fftconv_fwd(u, k, bias).float()
    expect = torch.zeros(B, D, L, device=dev)
    expect[:, :, p:] = k[:, : L - p]
    expect[:, :, p] += bias
    err = (out - expect).abs().max().item()
    tol = 2e-5 if dtype == torch.float32 else 2.0 ** -8 * expect.abs().max().item() + 1e-4
    assert err <= tol, err
    # (2) causality: changing u after position c does not change out before c
    u1 = rn(B, D, L).to(dtype)
    u2 = u1.clone()
    c = L // 3
    u2[:, :, c:] = rn(B, D, L - c).to(dtype)
    o1 = gpu_lib.fftconv_fwd(u1, k, bias).float()
    o2 = gpu_lib.fftconv_fwd(u2, k, bias).float()
    head = (o1[:, :, :c] - o2[:, :, :c]).abs().max().item()
    assert head <= (3e-6 if dtype == torch.float32 else 2.0 ** -7) * o1[:, :, :c].abs().max().item(), head
    # (3) adjoint identity <dout, conv(u)> = <du, u> = <dk, k> + <dbias, bias>  (fp32 accumulation on device)
    dout = rn(B, D, L).to(dtype)
    du, dk, dbias = gpu_lib.fftconv_bwd(dout, u1, k, bias)
    dot = lambda a, b: sum((a[i].double() * b[i].double()).sum().item() for i in range(a.shape[0]))   # noqa: E731
    lhs = dot(dout, o1)
    mid = dot(du, u1)
    rhs = dot(dk, k) + (dbias.double() * bias.double()).sum().item()
    scale = (dot(dout, dout) * dot(o1, o1)) ** 0.5
    rt = 2e-5 if dtype == torch.float32 else 6e-3
    assert abs(lhs - mid) <= rt * scale and abs(lhs - rhs) <= rt * scale, (lhs, mid, rhs, scale)
    # (4) determinism: same inputs, bitwise the same outputs (no atomics anywhere in the path)
    du_b, dk_b, dbias_b = gpu_lib.fftconv_bwd(dout, u1, k, bias)
    assert torch.equal(du, du_b) and torch.equal(dk, dk_b) and torch.equal(dbias, dbias_b)
    assert torch.isfinite(dk).all() and torch.isfinite(du.float()).all()


def test_streams_and_chunking_agree(gpu_lib):
    """Non-default stream + every chunk size give the same bits (the library uses the caller's stream)."""
    dev = torch.device("cuda", 0)
    u, k, bias, dout = _inputs(2, 12, 20000, torch.float32, seed=4)
    ud, kd, bd = u.to(dev), k.to(dev), bias.to(dev)
    ref = gpu_lib.fftconv_fwd(ud, kd, bd)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        outs = [gpu_lib.fftconv_fwd(ud, kd, bd, chunk=c) for c in (1, 5, 12)]
    s.synchronize()
    for o in outs:
        assert torch.equal(o, ref)


@pytest.mark.parametrize("B,D,L,dtype", [(2, 8, 32768, torch.bfloat16), (1, 2, 1048576, torch.float32), (3, 5, 1023, torch.float32)])
def test_saved_spectra_path_is_bitwise_the_recomputing_one(gpu_lib, B, D, L, dtype):
    dev = torch.device("cuda", 0)
    u, k, bias, dout = (t.to(dev) for t in _inputs(B, D, L, dtype, seed=L + 9))
    out = gpu_lib.fftconv_fwd(u, k, bias)
    du, dk, dbias = gpu_lib.fftconv_bwd(dout, u, k, bias)
    out2, saved = gpu_lib.fftconv_fwd(u, k, bias, save=True)
    # the workspace-free plan (L <= 32768) keeps the filter spectrum only and re-reads u; the two-level plan needs neither
    du2, dk2, dbias2 = gpu_lib.fftconv_bwd(dout, u if L <= 32768 else None, None, bias, saved=saved)
    torch.cuda.synchronize()
    assert torch.equal(out, out2) and torch.equal(du, du2) and torch.equal(dk, dk2) and torch.equal(dbias, dbias2)


def _ref_core(x, w, b, k, bias, L):
    D = x.shape[-1] // 3
    xc = O.short_conv(x.transpose(1, 2), w, b, L)
    x0, x1, v = xc.split(D, dim=1)
    return (O.fftconv_ref(v * x1, k, bias) * x0).transpose(1, 2)


@pytest.mark.parametrize("B,Lx,L,D,dtype", [(2, 70, 70, 8, torch.float32), (2, 2100, 2048, 70, torch.float32),
                                            (1, 130, 64, 64, torch.float32), (2, 5000, 5000, 256, torch.float32),
                                            (2, 3000, 3000, 128, torch.bfloat16)])
def test_fused_mixer_core_vs_oracle(gpu_lib, B, Lx, L, D, dtype):
    """short conv + gates + long conv + layout changes, fused HIP path, against the oracle pieces on the CPU."""
    from hyena_dna_amd.mixer import hyena_mixer_core
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(B * 1000 + L + D)
    x = torch.randn(B, Lx, 3 * D, generator=g).to(dtype)
    w = torch.randn(3 * D, 1, 3, generator=g) * 0.5
    b = torch.randn(3 * D, generator=g) * 0.1
    k = torch.randn(D, L, generator=g) * torch.exp(-4 * torch.linspace(0, 1, L)) * 0.2
    bias = torch.randn(D, generator=g)
    dz = torch.randn(B, L, D, generator=g).to(dtype)
    ts = [t.to(dev).requires_grad_(True) for t in (x, w, b, k, bias)]
    z = hyena_mixer_core(*ts, L)
    z.backward(dz.to(dev))
    got = [z.detach().cpu()] + [t.grad.cpu() for t in ts]
    rs = [t.clone().float().requires_grad_(True) for t in (x, w, b, k, bias)]
    zr = _ref_core(*rs, L)
    zr.backward(dz.float())
    ref = [zr.detach()] + [t.grad for t in rs]
    tol = 5e-6 if dtype == torch.float32 else 1.5e-2
    for n, a, r in zip(["z", "dx", "dw", "db", "dk", "dbias"], got, ref):
        assert a.shape == r.shape, n
        assert _rel(a.float(), r) < tol, (n, _rel(a.float(), r))
    # deterministic (no atomics in the short-filter gradient reduction)
    ts2 = [t.to(dev).requires_grad_(True) for t in (x, w, b, k, bias)]
    z2 = hyena_mixer_core(*ts2, L)
    z2.backward(dz.to(dev))
    assert torch.equal(z2, z) and all(torch.equal(a.grad, b_.grad) for a, b_ in zip(ts, ts2))


def test_hipgraph_capture_and_replay(gpu_lib):
    """The entry points allocate nothing, never synchronise and launch on the caller's stream, so forward + backward can
    be captured into a hipGraph (torch.cuda.CUDAGraph) and replayed on new data -- the launch-bound regime at L = 1k."""
    dev = torch.device("cuda", 0)
    u, k, bias, dout = (t.to(dev) for t in _inputs(4, 16, 1024, torch.bfloat16, seed=9))
    u2, k2, _, dout2 = (t.to(dev) for t in _inputs(4, 16, 1024, torch.bfloat16, seed=10))

    def run():
        out, saved = gpu_lib.fftconv_fwd(u, k, bias, save=True)
        return (out,) + tuple(gpu_lib.fftconv_bwd(dout, u, k, bias, saved=saved))

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                 # warm-up: twiddle tables (one synchronous copy) are created here
        run()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = run()
    for dst, src in ((u, u2), (k, k2), (dout, dout2)):
        dst.copy_(src)
    graph.replay()
    torch.cuda.synchronize()
    got = [t.clone() for t in captured]
    want = run()
    torch.cuda.synchronize()
    for a, b in zip(got, want):
        assert torch.equal(a, b)


def test_randomised_shapes_vs_oracle_gpu(gpu_lib):
    """24 seeded random (B, D, L, dtype, chunk, bias?) cases on the gfx950 binary, L up to 150000 (M1 = 1 ... 160)"""
    rng = torch.Generator().manual_seed(20240925)
    dts = [torch.float32, torch.bfloat16, torch.float16]
    lmax = [700, 5000, 40000, 150000]
    for case in range(24):
        B = int(torch.randint(1, 7, (1,), generator=rng))
        D = int(torch.randint(1, 7, (1,), generator=rng))
        L = int(torch.randint(1, lmax[case % 4] + 1, (1,), generator=rng))
        dtype = dts[int(torch.randint(0, 3, (1,), generator=rng))]
        chunk = int(torch.randint(0, D + 1, (1,), generator=rng))
        use_bias = bool(torch.randint(0, 4, (1,), generator=rng))
        u, k, bias, dout = _inputs(B, D, L, dtype, seed=2000 + case)
        if not use_bias:
            bias = torch.zeros(D)
        out, du, dk, dbias = _gpu(gpu_lib, u, k, bias, dout, chunk=chunk or None)
        r_out, r_du, r_dk, r_db = _oracle(u.float(), k, bias, dout.float())
        tag = (case, B, D, L, dtype, chunk, use_bias)
        if dtype == torch.float32:
            assert _rel(out, r_out) < 3e-6 and _rel(du, r_du) < 3e-6, tag
        else:
            tol = 2 ** -8 if dtype == torch.bfloat16 else 2 ** -11
            assert (out.float().cpu() - r_out).abs().max() <= tol * r_out.abs().max() + 1e-6, tag
            assert (du.float().cpu() - r_du).abs().max() <= tol * r_du.abs().max() + 1e-6, tag
        assert _rel(dk, r_dk) < 2e-5 and _rel(dbias, r_db) < 3e-5, tag


# ---- the BASELINE configurations themselves, every channel, element-wise against the oracle ---------------------
# (the one hardware bug of round 1 -- a buffer_store hazard -- only showed at D = 256; property checks are not
# a substitute for comparing the contract shapes with the oracle.  The CPU oracle needs ~20 s for the largest one.)
@pytest.mark.parametrize("B,D,L,dtype", [
    (8, 128, 1024, torch.float32),        # hyenadna-tiny-1k
    (8, 128, 1024, torch.bfloat16),
    (8, 256, 32768, torch.bfloat16),      # hyenadna-small-32k
    (2, 256, 160000, torch.bfloat16),     # hyenadna-medium-160k
    (1, 256, 450560, torch.bfloat16),     # hyenadna-medium-450k
    (1, 256, 1048576, torch.bfloat16),    # hyenadna-large-1m (the headline)
    (1, 256, 32767, torch.bfloat16),      # what the reference's dataset really yields: max_length - 1 (hg38_dataset.py:220)
])
def test_contract_configs_all_channels_vs_oracle(gpu_lib, B, D, L, dtype):
    u, k, bias, dout = _inputs(B, D, L, dtype, seed=L + D)
    out, du, dk, dbias = _gpu(gpu_lib, u, k, bias, dout)
    torch.set_num_threads(max(1, torch.get_num_threads()))
    r_out, r_du, r_dk, r_db = _oracle(u, k, bias, dout)
    if dtype == torch.float32:
        assert _rel(out, r_out) < REL_FP32 and _rel(du, r_du) < REL_FP32
    else:
        eps = 2.0 ** -7
        # forward: fp32 math and ONE rounding on both sides -> at most one bf16 ulp apart, bit-identical almost everywhere
        # (absolute slack: the fp32 noise floor of a transform whose row peaks at |r|max is ~1e-6 |r|max on BOTH sides, which
        # decides the rounding of the few samples that sit near zero -- 2.7e8 samples here)
        diff = (out.float() - r_out.float()).abs()
        slack = 2e-5 + 2e-6 * r_out.float().abs().amax(dim=2, keepdim=True)
        assert (diff <= eps * r_out.float().abs() + slack).all()
        assert (out != r_out).float().mean() < 0.02
        # per channel too: a defect confined to a few channels must not hide in a global average
        per_ch = (out != r_out).float().mean(dim=(0, 2))
        assert per_ch.max() < 0.04
        assert _rel(out.float(), r_out.float()) < 1e-3                # the north_star tolerance
        # du: the reference rounds the FFT branch and the bias branch separately (three roundings), we round once
        assert _rel(du.float(), r_du.float()) < 1.5 * eps
        ch = ((du.float() - r_du.float()).norm(dim=(0, 2)) / r_du.float().norm(dim=(0, 2))).max().item()
        assert ch < 1.5 * eps
    assert _rel(dk, r_dk) < REL_FP32
    ch_dk = ((dk.double() - r_dk.double()).norm(dim=1) / r_dk.double().norm(dim=1)).max().item()
    assert ch_dk < 2 * REL_FP32
    # dbias[d] is one sum of B L products (size ~ sqrt(B L), heavy cancellation): compare with the fp64 value on that scale
    db64 = (dout.double() * u.double()).sum(dim=(0, 2))
    assert (dbias.double() - db64).abs().max() < 3e-6 * (B * L) ** 0.5 + 1e-6       # ~3 fp32 ulps of sqrt(B L)


@pytest.mark.parametrize("B,D,L", [(4, 64, 4096), (2, 16, 32768), (1, 8, 100000)])
def test_entry_points_are_graph_capturable(gpu_lib, B, D, L):
    """include/hyena_fftconv.h promises no allocation / synchronisation inside the compute entry points: after one warm-up call
    (twiddle tables, workspace, LDS attributes) forward + backward of both plans are captured into a hipGraph and replayed
    on new data -- results equal the eager ones bit for bit."""
    dev = torch.device("cuda", 0)
    u, k, bias, dout = (t.to(dev) for t in _inputs(B, D, L, torch.bfloat16, seed=L))
    out_e = gpu_lib.fftconv_fwd(u, k, bias)                       # warm-up + eager reference
    du_e, dk_e, db_e = gpu_lib.fftconv_bwd(dout, u, k, bias)
    torch.cuda.synchronize()
    su, sd = torch.zeros_like(u), torch.zeros_like(dout)          # static inputs of the graph
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        gpu_lib.fftconv_fwd(su, k, bias)                          # warm-up on the capture stream (its own workspace)
        gpu_lib.fftconv_bwd(sd, su, k, bias)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            out_g = gpu_lib.fftconv_fwd(su, k, bias)
            du_g, dk_g, db_g = gpu_lib.fftconv_bwd(sd, su, k, bias)
    torch.cuda.current_stream().wait_stream(s)
    su.copy_(u)
    sd.copy_(dout)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out_g, out_e) and torch.equal(du_g, du_e) and torch.equal(dk_g, dk_e) and torch.equal(db_g, db_e)


def test_randomised_shapes_workspace_free_plan(gpu_lib):
    """60 seeded random (B, D, L, dtype, bias?) cases with L <= 32768 (every transform size 1024 ... 32768, ragged everything,
    channel counts with and without the XCD-aware row mapping) against the oracle"""
    rng = torch.Generator().manual_seed(20260924)
    dts = [torch.float32, torch.bfloat16, torch.float16]
    for case in range(60):
        r = int(torch.randint(0, 6, (1,), generator=rng))                       # transform size 1024 << r
        hi = 1024 << r
        L = int(torch.randint(hi // 2 + 1 if r else 1, hi + 1, (1,), generator=rng))
        B = int(torch.randint(1, 20, (1,), generator=rng))
        D = [1, 3, 8, 16, 24, 7][int(torch.randint(0, 6, (1,), generator=rng))]
        if B * D * L > 6_000_000:
            B = max(1, 6_000_000 // (D * L))
        dtype = dts[int(torch.randint(0, 3, (1,), generator=rng))]
        u, k, bias, dout = _inputs(B, D, L, dtype, seed=3000 + case)
        use_bias = bool(torch.randint(0, 4, (1,), generator=rng))
        dev = torch.device("cuda", 0)
        ud, kd, gd = u.to(dev), k.to(dev), dout.to(dev)
        bd = bias.to(dev) if use_bias else None
        out = gpu_lib.fftconv_fwd(ud, kd, bd)
        du, dk, dbias = gpu_lib.fftconv_bwd(gd, ud, kd, bd)
        r_out, r_du, r_dk, r_db = _oracle(u.float(), k, bias if use_bias else torch.zeros(D), dout.float())
        tag = (case, B, D, L, dtype, use_bias)
        tol = REL_FP32 if dtype == torch.float32 else (6e-3 if dtype == torch.bfloat16 else 8e-4)
        assert _rel(out.float(), r_out) < tol and _rel(du.float(), r_du) < tol, tag
        assert _rel(dk, r_dk) < REL_FP32, tag
        db64 = (dout.double() * u.double()).sum(dim=(0, 2))
        assert (dbias.double().cpu() - db64).abs().max() < 3e-6 * (B * L) ** 0.5 + 1e-5, tag


@pytest.mark.parametrize("B,D,L,dtype", [
    (3, 8, 32767, torch.bfloat16), (2, 8, 159999, torch.bfloat16), (1, 8, 449999, torch.bfloat16), (1, 8, 999999, torch.bfloat16),
    (1, 8, 1048575, torch.bfloat16), (2, 3, 32767, torch.float32), (1, 2, 1048575, torch.float32), (2, 5, 1023, torch.float16),
    # every channel of the model width at the two odd lengths no other test reaches at D = 256 (round 1's only hardware bug
    # showed at D = 256 and nowhere else)
    (1, 256, 449999, torch.bfloat16), (1, 256, 999999, torch.bfloat16),
])
def test_real_training_lengths_on_pitched_rows(gpu_lib, B, D, L, dtype):
    """L = max_length - 1, the only lengths the reference's trainer produces (hg38_dataset.py:220-223: 32 767, 159 999, 449 999, 999 999,
    1 048 575): the operator hands the convolution PITCHED rows there (_lib.empty_rows: rows 64 elements apart, hyena_fftconv_fwd_ld /
    _bwd_ld).  Same arithmetic as on packed rows, so the results must be the packed call's BITS -- and both the oracle's values."""
    dev = torch.device("cuda", 0)
    u, k, bias, dout = _inputs(B, D, L, dtype, seed=L)

    def pitched(t):
        r = gpu_lib.empty_rows(t.shape[:-1], t.shape[-1], t.dtype, dev)
        r.copy_(t)
        return r

    assert gpu_lib.row_pitch(L) % 64 == 0 and gpu_lib.row_pitch(L) - L < 64
    res = {}
    for name, lay in (("packed", lambda t: t.to(dev)), ("pitched", pitched)):
        ud, kd, gd = lay(u), lay(k), lay(dout)
        bd = bias.to(dev)
        if name == "pitched":
            assert gpu_lib.ld_of(ud) == gpu_lib.row_pitch(L) and not ud.is_contiguous() and ud.data_ptr() % 128 == 0       # (B D > 1 in every case)
            # poison what lies between the rows: nothing may read it, and nothing may write it
            for t in (ud, kd, gd):
                buf = torch.as_strided(t, t.shape[:-1] + (gpu_lib.ld_of(t),), t.stride())
                buf[..., L:] = float("nan")
        out, saved = gpu_lib.fftconv_fwd(ud, kd, bd, save=True)
        du, dk, dbias = gpu_lib.fftconv_bwd(gd, ud, kd, bd, saved=saved)
        du2, dk2, dbias2 = gpu_lib.fftconv_bwd(gd, ud, kd, bd)                  # the recomputing backward reads u and k again
        assert torch.equal(du, du2) and torch.equal(dk, dk2) and torch.equal(dbias, dbias2)
        if name == "pitched":
            assert gpu_lib.ld_of(out) == gpu_lib.ld_of(ud) and gpu_lib.ld_of(du) == gpu_lib.ld_of(ud) and gpu_lib.ld_of(dk) == gpu_lib.ld_of(kd)
        res[name] = [t.cpu() for t in (out, du, dk, dbias)]
        assert all(torch.isfinite(t).all() for t in res[name])
    for a, b in zip(res["packed"], res["pitched"]):
        assert torch.equal(a, b)
    out, du, dk, dbias = res["pitched"]
    r_out, r_du, r_dk, r_db = _oracle(u.float(), k, bias, dout.float())
    tol = REL_FP32 if dtype == torch.float32 else (6e-3 if dtype == torch.bfloat16 else 8e-4)
    assert _rel(out.float(), r_out) < tol and _rel(du.float(), r_du) < tol
    assert _rel(dk, r_dk) < REL_FP32
    db64 = (dout.double() * u.double()).sum(dim=(0, 2))
    assert (dbias.double() - db64).abs().max() < 3e-6 * (B * L) ** 0.5 + 1e-5
    if D >= 64:                               # per channel: one bad channel must not hide in the global norm
        ch = ((out.float() - r_out.float()).norm(dim=(0, 2)) / r_out.float().norm(dim=(0, 2))).max().item()
        ch_du = ((du.float() - r_du.float()).norm(dim=(0, 2)) / r_du.float().norm(dim=(0, 2))).max().item()
        ch_dk = ((dk.double() - r_dk.double()).norm(dim=1) / r_dk.double().norm(dim=1)).max().item()
        assert ch < 1.5 * tol and ch_du < 1.5 * tol and ch_dk < 2 * REL_FP32, (ch, ch_du, ch_dk)
