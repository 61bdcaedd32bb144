"""The HIP kernel sources (hyena_dna_amd/csrc) executed under tests/hipemu on the CPU, through the C ABI, against
the oracle and the golden vectors minted from the reference.  This checks the algorithm and every index map in
the build container; the `-m gpu` tests check the same entry points compiled for gfx950 on a real MI355X."""
import pytest
import torch

from oracle import hyena_oracle as O

REL_FP32 = 2e-6        # rel-L2 vs the reference fp32 path (which itself is ~1e-6 from the fp64 truth)


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def _oracle(u, k, bias, dout):
    u_ = u.clone().requires_grad_(True)
    k_ = k.clone().requires_grad_(True)
    b_ = bias.clone().requires_grad_(True)
    out = O.fftconv_ref(u_, k_, b_)
    out.backward(dout)
    return out.detach(), u_.grad, k_.grad, b_.grad


def _inputs(B, D, L, dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    u = torch.randn(B, D, L, generator=g).to(dtype)
    k = torch.randn(D, L, generator=g) * torch.exp(-5.0 * torch.linspace(0, 1, L))[None] * 0.1
    bias = torch.randn(D, generator=g)
    dout = torch.randn(B, D, L, generator=g).to(dtype)
    return u, k, bias, dout


@pytest.mark.parametrize("B,D,L,chunk", [
    (2, 3, 1, 0), (2, 3, 8, 0), (1, 2, 1023, 0), (2, 2, 1024, 1), (1, 3, 1025, 2), (1, 2, 2048, 0),
    (2, 2, 3000, 0), (1, 2, 8191, 0), (1, 1, 16384, 0), (2, 2, 32768, 1), (1, 1, 65536, 0), (1, 2, 100000, 0),
    (1, 1, 160000, 0), (2, 2, 131073, 1), (1, 2, 163839, 0), (1, 1, 163840, 0), (1, 1, 163841, 0), (1, 1, 262144, 0),
    (1, 1, 450560, 0), (2, 1, 262145, 0), (1, 1, 460800, 0), (4, 2, 5000, 0), (5, 3, 9000, 2), (8, 1, 40000, 0),
])
def test_fp32_fwd_bwd_vs_oracle(emu_backend, B, D, L, chunk):
    u, k, bias, dout = _inputs(B, D, L, torch.float32, seed=L)
    out = emu_backend.fftconv_fwd(u, k, bias, chunk=chunk)
    du, dk, dbias = emu_backend.fftconv_bwd(dout, u, k, bias, chunk=chunk)
    r_out, r_du, r_dk, r_db = _oracle(u, k, bias, dout)
    assert _rel(out, r_out) < REL_FP32
    assert _rel(du, r_du) < REL_FP32
    assert _rel(dk, r_dk) < REL_FP32
    # dbias is ONE fp32 sum of B*L products per channel; the fp32 oracle itself is ~1e-5 from the exact value at L > 1e5
    assert _rel(dbias, r_db) < (5e-6 if L <= 100000 else 2e-5)


def test_fp32_L_1M(emu_backend):
    """The headline length: one row, N = 2^21 (M1 = 1024), odd and even L."""
    for L in (1048576, 1048575):
        u, k, bias, dout = _inputs(1, 1, L, torch.float32, seed=7)
        out = emu_backend.fftconv_fwd(u, k, bias)
        du, dk, dbias = emu_backend.fftconv_bwd(dout, u, k, bias)
        r_out, r_du, r_dk, r_db = _oracle(u, k, bias, dout)
        assert _rel(out, r_out) < 3e-6 and _rel(du, r_du) < 3e-6 and _rel(dk, r_dk) < 3e-6
        assert _rel(dbias, r_db) < 1e-5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("L", [37, 1000, 1023, 4100, 40000])
def test_half_io_matches_reference_rounding(emu_backend, dtype, L):
    """16-bit I/O: identical inputs, fp32 math inside, one rounding at the end -- so we must agree with the
    reference to within one 16-bit ulp almost everywhere (an fp32 difference of ~1e-6 flips a rounding rarely)."""
    u, k, bias, dout = _inputs(2, 3, L, dtype, seed=L + 1)
    out = emu_backend.fftconv_fwd(u, k, bias)
    du, dk, dbias = emu_backend.fftconv_bwd(dout, u, k, bias)
    r_out, r_du, r_dk, r_db = _oracle(u, k, bias, dout)
    assert out.dtype == dtype and du.dtype == dtype and dk.dtype == torch.float32
    eps = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10     # one ulp of the 16-bit format
    # forward: one rounding on both sides -> at most one ulp apart, and bit-identical almost everywhere
    diff = (out.float() - r_out.float()).abs()
    assert (diff <= eps * r_out.float().abs() + 2e-5).all()
    assert (out != r_out).float().mean() < 0.02
    # du: the reference's autograd rounds the FFT branch and the bias branch to 16 bits separately and adds them
    # in 16 bits (three roundings); we round once.  So compare both with the fp32 result on the same 16-bit
    # inputs: we must be within half an ulp of it (correctly rounded); the reference is only norm-wise close.
    t_out, t_du, _, _ = _oracle(u.float(), k, bias, dout.float())
    assert ((du.float() - t_du).abs() <= 0.5 * eps * t_du.abs() * 1.001 + 2e-5).all()
    assert ((out.float() - t_out).abs() <= 0.5 * eps * t_out.abs() * 1.001 + 2e-5).all()
    assert _rel(du.float(), r_du.float()) < 1.5 * eps       # norm-wise only: its error scales with the branches
    assert _rel(dk, r_dk) < REL_FP32 and _rel(dbias, r_db) < 5e-6


@pytest.mark.parametrize("name", ["b1d2l40000", "b1d1l160000_bf16"])
def test_golden_vectors_from_reference_large(emu_backend, golden_fftconv_large, name):
    """reference-minted vectors beyond one column: L = 40000 (two-stage columns, M1 = 64), L = 160000 (mixed radix, M1 = 160)"""
    c = golden_fftconv_large[name]
    out = emu_backend.fftconv_fwd(c["u"], c["k"], c["bias"])
    du, dk, dbias = emu_backend.fftconv_bwd(c["dout"], c["u"], c["k"], c["bias"])
    if c["u"].dtype == torch.float32:
        assert _rel(out, c["out"]) < REL_FP32 and _rel(du, c["du"]) < REL_FP32
    else:
        eps = 2.0 ** -7
        diff = (out.float() - c["out"].float()).abs()
        assert (diff <= eps * c["out"].float().abs() + 2e-5).all() and (out != c["out"]).float().mean() < 0.02
        assert _rel(du.float(), c["du"].float()) < 1.5 * eps
    assert _rel(dk, c["dk"]) < REL_FP32 and _rel(dbias, c["dbias"]) < 2e-5


@pytest.mark.parametrize("name", ["b2d4l8", "b2d3l37", "b1d4l1023", "b2d4l1024", "b2d4l1024_5d", "b2d4l1000_bf16",
                                  "b1d2l4100"])
def test_golden_vectors_from_reference(emu_backend, golden_fftconv, name):
    c = golden_fftconv[name]
    out = emu_backend.fftconv_fwd(c["u"], c["k"], c["bias"])
    du, dk, dbias = emu_backend.fftconv_bwd(c["dout"], c["u"], c["k"], c["bias"])
    if c["u"].dtype == torch.float32:
        assert _rel(out, c["out"]) < REL_FP32 and _rel(du, c["du"]) < REL_FP32
    else:
        assert (out.float() - c["out"].float()).abs().max() <= 2.0 ** -7 * c["out"].float().abs().max()
        assert (du.float() - c["du"].float()).abs().max() <= 2.0 ** -7 * c["du"].float().abs().max()
    assert _rel(dk, c["dk"]) < REL_FP32
    assert _rel(dbias, c["dbias"]) < 5e-6


def test_no_bias_and_partial_grads(emu_backend):
    u, k, bias, dout = _inputs(2, 2, 777, torch.float32, seed=3)
    out = emu_backend.fftconv_fwd(u, k, None)
    assert _rel(out, O.fftconv_ref(u, k, torch.zeros(2))) < REL_FP32
    du, dk, dbias = emu_backend.fftconv_bwd(dout, u, k, bias, need_du=True, need_dk=False)
    assert dk is None and dbias is None
    du2, dk2, _ = emu_backend.fftconv_bwd(dout, u, k, bias, need_du=False, need_dk=True)
    assert du2 is None
    _, r_du, r_dk, _ = _oracle(u, k, bias, dout)
    assert _rel(du, r_du) < REL_FP32 and _rel(dk2, r_dk) < REL_FP32


def test_error_codes(emu_backend):
    L = emu_backend.lib()
    assert L.hyena_fftconv_fft_size(0) == 0 and L.hyena_fftconv_fft_size(1048577) == 0
    assert L.hyena_fftconv_fft_size(1) == 1024 and L.hyena_fftconv_fft_size(1025) == 2048
    assert L.hyena_fftconv_fft_size(1048576) == 1048576
    u, k, bias, _ = _inputs(1, 2, 64, torch.float32)
    out = torch.empty_like(u)
    t = emu_backend.tables_for(u.device, 64)
    ws = torch.empty(16, dtype=torch.uint8)
    st = L.hyena_fftconv_fwd(u.data_ptr(), k.data_ptr(), None, out.data_ptr(), 1, 2, 64, 0, t.data_ptr(),
                             ws.data_ptr(), ws.numel(), 0, None)
    assert st == 3 and b"workspace" in L.hyena_fftconv_error_string(st)
    st = L.hyena_fftconv_fwd(None, k.data_ptr(), None, out.data_ptr(), 1, 2, 64, 0, t.data_ptr(), ws.data_ptr(),
                             ws.numel(), 0, None)
    assert st == 1
    st = L.hyena_fftconv_fwd(u.data_ptr(), k.data_ptr(), None, out.data_ptr(), 1, 2, 64, 7, t.data_ptr(),
                             ws.data_ptr(), ws.numel(), 0, None)
    assert st == 1


@pytest.mark.parametrize("B,D,L,chunk,dtype", [(1, 3, 5000, 2, torch.float32), (2, 4, 1023, 0, torch.bfloat16),
                                               (3, 2, 40000, 1, torch.float32), (1, 1, 262144, 0, torch.float32)])
def test_saved_spectra_backward_is_bitwise_the_recomputing_one(emu_backend, B, D, L, chunk, dtype):
    """fwd_save + bwd_saved (spectra kept from the forward) give the same bits as fwd + bwd (recomputed)."""
    u, k, bias, dout = _inputs(B, D, L, dtype, seed=L + 5)
    out = emu_backend.fftconv_fwd(u, k, bias, chunk=chunk)
    du, dk, dbias = emu_backend.fftconv_bwd(dout, u, k, bias, chunk=chunk)
    out2, saved = emu_backend.fftconv_fwd(u, k, bias, chunk=chunk, save=True)
    assert saved.numel() == emu_backend.saved_bytes(B, D, L)
    # the workspace-free plan (L <= 32768) saves the filter spectrum only and re-reads u; the two-level plan needs neither
    u_arg = u if L <= 32768 else None
    du2, dk2, dbias2 = emu_backend.fftconv_bwd(dout, u_arg, None, bias, chunk=chunk, saved=saved)
    assert torch.equal(out, out2) and torch.equal(du, du2) and torch.equal(dk, dk2) and torch.equal(dbias, dbias2)
    du3, dk3, _ = emu_backend.fftconv_bwd(dout, None, None, bias, need_du=True, need_dk=False, chunk=chunk, saved=saved)
    assert dk3 is None and torch.equal(du3, du)


def test_randomised_shapes_vs_oracle(emu_backend):
    """40 seeded random (B, D, L, dtype, chunk, bias?) cases, L up to 6000 (M1 = 1 ... 8), ragged everything"""
    rng = torch.Generator().manual_seed(20240924)
    dts = [torch.float32, torch.bfloat16, torch.float16]
    for case in range(40):
        B = int(torch.randint(1, 7, (1,), generator=rng))
        D = int(torch.randint(1, 6, (1,), generator=rng))
        L = int(torch.randint(1, 6001, (1,), generator=rng))
        dtype = dts[int(torch.randint(0, 3, (1,), generator=rng))]
        chunk = int(torch.randint(0, D + 1, (1,), generator=rng))
        u, k, bias, dout = _inputs(B, D, L, dtype, seed=1000 + case)
        use_bias = bool(torch.randint(0, 4, (1,), generator=rng))
        bb = bias if use_bias else None
        out = emu_backend.fftconv_fwd(u, k, bb, chunk=chunk)
        du, dk, dbias = emu_backend.fftconv_bwd(dout, u, k, bb, chunk=chunk)
        r_out, r_du, r_dk, r_db = _oracle(u.float(), k, bias if use_bias else torch.zeros(D), dout.float())
        tag = (case, B, D, L, dtype, chunk, use_bias)
        if dtype == torch.float32:
            assert _rel(out, r_out) < REL_FP32 and _rel(du, r_du) < REL_FP32, tag
        else:   # one 16-bit rounding of the fp32 result
            tol = 2 ** -8 if dtype == torch.bfloat16 else 2 ** -11
            assert (out.float() - r_out).abs().max() <= tol * r_out.abs().max() + 1e-6, tag
            assert (du.float() - r_du).abs().max() <= tol * r_du.abs().max() + 1e-6, tag
        assert _rel(dk, r_dk) < 2e-5 if dtype != torch.float32 else _rel(dk, r_dk) < REL_FP32, tag
        assert _rel(dbias, r_db) < 2e-5, tag


@pytest.mark.parametrize("m1", [3, 5, 6, 7, 10, 12, 14, 20, 24, 28, 96, 192, 224, 320, 384])
def test_mixed_radix_column_sizes(emu_backend, monkeypatch, m1):
    """every column size that is not a power of two (2^a x {3, 5, 7}; odd M1 has no self-paired row M1/2), forward + backward
    vs the fp64 evaluation of the oracle (the fp32 oracle itself is up to 5e-6 off at these non-smooth 2L)"""
    monkeypatch.setenv("HYENA_FFTCONV_ONCHIP", "0")        # M1 < 32: only the two-level plan has these sizes
    L = m1 * 1024 - 3
    assert emu_backend.lib().hyena_fftconv_fft_size(L) == m1 * 1024
    B, D = (2, 2) if m1 < 32 else (1, 1)
    u, k, bias, dout = _inputs(B, D, L, torch.float32, seed=L)
    out = emu_backend.fftconv_fwd(u, k, bias)
    du, dk, dbias = emu_backend.fftconv_bwd(dout, u, k, bias)
    r_out, r_du, r_dk, r_db = _oracle(u.double(), k.double(), bias.double(), dout.double())
    assert _rel(out, r_out) < 1e-6 and _rel(du, r_du) < 1e-6 and _rel(dk, r_dk) < 1e-6
    assert _rel(dbias, r_db) < 2e-5
