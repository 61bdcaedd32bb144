"""The workspace-free plan (hyena_dna_amd/csrc/onchip_kernels.h, L <= 32768) executed under tests/hipemu, through the C ABI,
against the oracle: every transform size 1024 R, ragged lengths, batch counts around the dk kernel's group count, channel
counts with and without the XCD-aware row mapping, all element types -- and bitwise agreement of what must be bitwise."""
import pytest
import torch

from oracle import hyena_oracle as O


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


# (B, D, L): R = 1 ... 32; B around the dk kernel's row-group counts (16, 8, 4, 2, 1); D = 8 / 16 take the XCD-aware mapping
CASES = [
    (1, 1, 1), (3, 2, 2), (17, 3, 700), (16, 8, 1024), (5, 1, 1023), (9, 2, 1025), (8, 8, 2048), (3, 3, 2047),
    (5, 2, 4096), (4, 1, 4095), (2, 16, 3000), (3, 2, 8192), (2, 1, 8191), (1, 3, 5000), (2, 2, 16384), (3, 1, 16383),
    (1, 2, 12000), (2, 2, 32768), (1, 1, 32767), (3, 1, 16385), (2, 1, 20000), (1, 8, 32768),
    # D far below the CU count: dk cuts the batch of a channel into slices (3 / 5 / 4 / 5 of them here, the last one ragged)
    (40, 3, 700), (9, 4, 5000), (7, 2, 2100), (5, 2, 20000),
]


@pytest.mark.parametrize("B,D,L", CASES)
def test_fp32_vs_oracle(emu_backend, B, D, L):
    assert emu_backend.lib().hyena_fftconv_plan(L) == emu_backend.PLAN_ONCHIP
    u, k, bias, dout = _inputs(B, D, L, torch.float32, seed=L + B)
    out = emu_backend.fftconv_fwd(u, k, bias)
    du, dk, dbias = emu_backend.fftconv_bwd(dout, u, k, bias)
    r_out, r_du, r_dk, r_db = _oracle(u, k, bias, dout)
    assert _rel(out, r_out) < 2e-6 and _rel(du, r_du) < 2e-6 and _rel(dk, r_dk) < 2e-6
    # dbias[d] is one sum of B L products of unit-variance numbers (typical size sqrt(B L), heavy cancellation): compare with
    # the fp64 value on that scale (the fp32 oracle's own summation error is of the same order)
    db64 = (dout.double() * u.double()).sum(dim=(0, 2))
    assert (dbias.double() - db64).abs().max() < 1e-6 * (B * L) ** 0.5 + 1e-6
    # options: no bias, du only, dk only -- the same bits as the full call
    du2, dk2, db2 = emu_backend.fftconv_bwd(dout, u, k, bias, need_du=True, need_dk=False)
    assert dk2 is None and db2 is None and torch.equal(du2, du)
    du3, dk3, db3 = emu_backend.fftconv_bwd(dout, u, k, bias, need_du=False, need_dk=True)
    assert du3 is None and torch.equal(dk3, dk) and torch.equal(db3, dbias)
    out0 = emu_backend.fftconv_fwd(u, k, None)
    assert _rel(out0, O.fftconv_ref(u, k, torch.zeros(D))) < 2e-6


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,D,L", [(3, 2, 37), (2, 3, 1000), (4, 2, 4100), (2, 1, 9000), (1, 2, 32767)])
def test_half_io(emu_backend, dtype, B, D, L):
    """16-bit I/O: fp32 math inside, one rounding on store -> within one ulp of the reference, bit-identical almost everywhere."""
    u, k, bias, dout = _inputs(B, D, L, dtype, seed=L + 1)
    out = emu_backend.fftconv_fwd(u, k, bias)
    du, dk, dbias = emu_backend.fftconv_bwd(dout, u, k, bias)
    r_out, r_du, r_dk, r_db = _oracle(u, k, bias, dout)
    assert out.dtype == dtype and du.dtype == dtype and dk.dtype == torch.float32
    eps = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    diff = (out.float() - r_out.float()).abs()
    assert (diff <= eps * r_out.float().abs() + 2e-5).all()
    assert (out != r_out).float().mean() < 0.02
    # du against the fp32 result on the same 16-bit inputs (the reference rounds three times, we once)
    _, f_du, f_dk, f_db = _oracle(u.float(), k, bias, dout.float())
    assert ((du.float() - f_du).abs() <= 0.5 * eps * f_du.abs() * 1.01 + 2e-5).all()
    assert _rel(dk, f_dk) < 2e-6
    assert (dbias.double() - (dout.double() * u.double()).sum(dim=(0, 2))).abs().max() < 1e-6 * (B * L) ** 0.5 + 1e-6


def test_same_results_from_both_plans(emu_backend, monkeypatch):
    """L <= 32768 on the two-level plan (HYENA_FFTCONV_ONCHIP=0) and on the workspace-free plan: two implementations of one
    operator, both within fp32 tolerance of the oracle and of each other."""
    for (B, D, L) in [(2, 3, 1000), (2, 2, 5000), (1, 2, 20000)]:
        u, k, bias, dout = _inputs(B, D, L, torch.float32, seed=L)
        monkeypatch.setenv("HYENA_FFTCONV_ONCHIP", "1")
        out_a = emu_backend.fftconv_fwd(u, k, bias)
        g_a = emu_backend.fftconv_bwd(dout, u, k, bias)
        monkeypatch.setenv("HYENA_FFTCONV_ONCHIP", "0")
        assert emu_backend.lib().hyena_fftconv_plan(L) == emu_backend.PLAN_TWO_LEVEL
        out_b = emu_backend.fftconv_fwd(u, k, bias)
        g_b = emu_backend.fftconv_bwd(dout, u, k, bias)
        assert _rel(out_a, out_b) < 1e-6
        for a, b in zip(g_a, g_b):
            assert _rel(a, b) < 2e-6


def test_properties(emu_backend):
    """Size-independent properties: impulse response = filter (+ bias at lag 0), causality, adjoint identities, determinism."""
    B, D, L = 2, 2, 6000
    u, k, bias, dout = _inputs(B, D, L, torch.float32, seed=3)
    imp = torch.zeros(B, D, L)
    imp[:, :, 0] = 1.0
    out = emu_backend.fftconv_fwd(imp, k, bias)
    ref = k.clone()
    ref[:, 0] += bias
    assert (out - ref[None]).abs().max() < 2e-6
    # causality: changing u from position p on leaves out[:p] unchanged (up to fp32 noise of the transform)
    p = 4321
    u2 = u.clone()
    u2[:, :, p:] = torch.randn(B, D, L - p)
    o1, o2 = emu_backend.fftconv_fwd(u, k, bias), emu_backend.fftconv_fwd(u2, k, bias)
    assert (o1[:, :, :p] - o2[:, :, :p]).abs().max() < 5e-5
    # adjoint: <dout, conv(u)> = <du, u> = <dk, k> + <dbias, bias>
    du, dk, dbias = emu_backend.fftconv_bwd(dout, u, k, bias)
    lhs = (dout.double() * o1.double()).sum()
    assert abs(lhs - (du.double() * u.double()).sum()) < 1e-6 * abs(lhs) + 1e-3
    assert abs(lhs - ((dk.double() * k.double()).sum() + (dbias.double() * bias.double()).sum())) < 1e-6 * abs(lhs) + 1e-3
    du_b, dk_b, dbias_b = emu_backend.fftconv_bwd(dout, u, k, bias)
    assert torch.equal(du, du_b) and torch.equal(dk, dk_b) and torch.equal(dbias, dbias_b)


def test_dk_batch_slices_cover_the_batch_exactly():
    """host logic of the sliced dk (csrc/onchip.hip dk_slices), read back through the C ABI: the backward's workspace is the filter
    spectrum + S D L floats of partial rows (S = 1: none) (+ at B = 1, M = 32768 the spectrum of u that dk's spectrum-plus-conjugate-
    convolution form uses).  A channel count >= the CU count is never sliced, the slice count never exceeds the sequential steps an
    unsliced workgroup would take, and D S stays within ~2x the CUs."""
    from hyena_dna_amd import _lib
    L_ = _lib.lib()
    for L, R in ((700, 1), (2000, 2), (4000, 4), (8000, 8), (16000, 16), (32000, 32)):
        bp = 1 if R == 32 else min(16, 512 // (32 * R))
        for B in (1, 2, 7, 8, 16, 17, 64, 250):
            for D in (1, 3, 64, 128, 200, 256, 768):
                extra = L_.hyena_fftconv_workspace_bytes(B, D, L, 1, 0) - L_.hyena_fftconv_workspace_bytes(B, D, L, 0, 0)
                if B == 1 and R == 32:
                    extra -= D * 1024 * R * 8                # the u spectrum [D][M] complex64
                assert extra % (D * L * 4) == 0
                S = extra // (D * L * 4) or 1
                assert extra == (S * D * L * 4 if S > 1 else 0)
                steps = -(-B // bp)
                assert 1 <= S <= steps and (S == 1 if D >= 256 else S * D < 2 * 256 + D)
                if D < 256 and steps >= 2:
                    assert S >= 2                        # the case this exists for


# ---- short rows, small batch: one launch per direction (small_fwd_kernel / small_bwd_kernel) ------------------------------
SMALL_CASES = [(8, 5, 1024), (3, 3, 1000), (16, 2, 700), (1, 2, 37), (5, 3, 2048), (8, 2, 1500), (2, 9, 1), (7, 1, 1025)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,D,L", SMALL_CASES)
def test_small_fused_pair_vs_oracle(emu_backend, monkeypatch, B, D, L, dtype):
    """The fused forward keeps H for the fused backward (saved spectra); results vs the oracle, vs the general kernels of the
    same library (HYENA_FFTCONV_SMALL=0: same transform, another summation order over the batch), options bitwise."""
    u, k, bias, dout = _inputs(B, D, L, dtype, seed=3 * L + B)
    out, saved = emu_backend.fftconv_fwd(u, k, bias, save=True)
    du, dk, dbias = emu_backend.fftconv_bwd(dout, u, k, bias, saved=saved)
    assert torch.equal(out, emu_backend.fftconv_fwd(u, k, bias))                 # with or without leaving H behind
    monkeypatch.setenv("HYENA_FFTCONV_SMALL", "0")
    g_out, g_saved = emu_backend.fftconv_fwd(u, k, bias, save=True)
    g_du, g_dk, g_db = emu_backend.fftconv_bwd(dout, u, k, bias, saved=g_saved)
    monkeypatch.delenv("HYENA_FFTCONV_SMALL")
    assert torch.equal(out, g_out) and torch.equal(du, g_du)                       # row by row the same arithmetic
    assert torch.equal(saved.view(torch.float32), g_saved.view(torch.float32))
    assert _rel(dk, g_dk) < 1e-6 and (dbias - g_db).abs().max() < 1e-5 * (B * L) ** 0.5
    r_out, r_du, r_dk, r_db = _oracle(u.float(), k, bias, dout.float())
    tol = 2e-6 if dtype == torch.float32 else (2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10)
    assert _rel(out.float(), r_out) < tol and _rel(du.float(), r_du) < tol and _rel(dk, r_dk) < 2e-6
    db64 = (dout.double() * u.double()).sum(dim=(0, 2))
    assert (dbias.double() - db64).abs().max() < 1e-6 * (B * L) ** 0.5 + 1e-6
    # du only / dk only: the same bits as the combined launch; and run to run
    du2, dk2, db2 = emu_backend.fftconv_bwd(dout, u, k, bias, need_du=True, need_dk=False, saved=saved)
    assert dk2 is None and db2 is None and torch.equal(du2, du)
    du3, dk3, db3 = emu_backend.fftconv_bwd(dout, u, k, bias, need_du=False, need_dk=True, saved=saved)
    assert du3 is None and torch.equal(dk3, dk) and torch.equal(db3, dbias)
    du4, dk4, db4 = emu_backend.fftconv_bwd(dout, u, k, bias, saved=saved)
    assert torch.equal(du4, du) and torch.equal(dk4, dk) and torch.equal(db4, dbias)
    # without the forward's spectrum: the filter is transformed by the same code first -- the same bits
    du5, dk5, db5 = emu_backend.fftconv_bwd(dout, u, k, bias)
    assert torch.equal(du5, du) and torch.equal(dk5, dk) and torch.equal(db5, dbias)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("D,L", [(3, 3000), (2, 8000), (9, 12000), (2, 32768), (1, 20001)])
def test_dk_at_batch_one_is_spectrum_plus_conjugate_conv(emu_backend, monkeypatch, D, L, dtype):
    """B = 1, M = 32768: dk = corr(dout, u) runs as the forward's two kernels (spectrum of u, convolution with the conjugate, fp32
    rows) instead of dk_kernel's two parity launches; HYENA_FFTCONV_DK1=0 keeps dk_kernel -- both against the oracle, du untouched by
    the choice (the shorter rows: one code path either way)"""
    u, k, bias, dout = _inputs(1, D, L, dtype, seed=L + D)
    du, dk, dbias = emu_backend.fftconv_bwd(dout, u, k, bias)
    monkeypatch.setenv("HYENA_FFTCONV_DK1", "0")
    du0, dk0, dbias0 = emu_backend.fftconv_bwd(dout, u, k, bias)
    monkeypatch.delenv("HYENA_FFTCONV_DK1")
    _, _, r_dk, r_db = _oracle(u.float(), k, bias, dout.float())
    assert torch.equal(du, du0)
    assert dk.dtype == torch.float32 and _rel(dk, r_dk) < 2e-6 and _rel(dk0, r_dk) < 2e-6
    assert torch.equal(dbias, dk[:, 0]) and (dbias - r_db).abs().max() < 1e-5 * L ** 0.5 + 1e-5
    du1, dk1, db1 = emu_backend.fftconv_bwd(dout, u, k, bias, need_du=False, need_dk=True)
    assert du1 is None and torch.equal(dk1, dk) and torch.equal(db1, dbias)


# (B, D, L, dtype): odd lengths of every kernel family -- one-launch pair (R <= 2, small batch), general conv / dk (with and without batch
# slices, T = 32 with its whole-tensor descriptor), dk at B = 1 as spectrum + conjugate conv, the two parity launches at M = 32768,
# and the two-level plan
PITCHED_CASES = [
    (2, 3, 999, torch.float32), (3, 2, 2047, torch.bfloat16), (17, 3, 701, torch.float32), (40, 3, 703, torch.bfloat16),
    (9, 2, 1025, torch.float16), (3, 2, 4095, torch.bfloat16), (1, 2, 8191, torch.float32), (2, 1, 16383, torch.bfloat16),
    (2, 2, 32767, torch.bfloat16), (1, 2, 32767, torch.bfloat16), (1, 1, 32767, torch.float32), (2, 2, 40001, torch.bfloat16),
    (1, 3, 33333, torch.float32),
]


@pytest.mark.parametrize("B,D,L,dtype", PITCHED_CASES)
def test_pitched_rows_give_the_packed_bits(emu_backend, B, D, L, dtype):
    """Round 5: hyena_fftconv_fwd_ld / _bwd_ld on rows that are row_pitch(L) elements apart (what the operator's fused path hands the
    convolution at the reference trainer's odd lengths) -- the same arithmetic on the same values, so the packed call's bits; what lies
    between the rows is poisoned with NaN on the input side and must come back untouched on the output side."""
    _lib = emu_backend
    u, k, bias, dout = _inputs(B, D, L, dtype, seed=L + B)
    ld = _lib.row_pitch(L)
    assert ld % 64 == 0 and 0 < ld - L < 64

    def pitched(t, fill):
        buf = torch.full(t.shape[:-1] + (ld,), fill, dtype=t.dtype)
        buf[..., :L] = t
        return buf[..., :L], buf

    ref_out, ref_saved = _lib.fftconv_fwd(u, k, bias, save=True)
    ref = [ref_out] + list(_lib.fftconv_bwd(dout, u, k, bias, saved=ref_saved))
    (up, _), (kp, _), (gp, _) = pitched(u, float("nan")), pitched(k, float("nan")), pitched(dout, float("nan"))
    one = lambda t: t.numel() == t.shape[-1]                                  # a single row has no pitch to speak of: ld_of says L
    assert _lib.ld_of(up) == (L if one(up) else ld) and _lib.ld_of(kp) == (L if one(kp) else ld)
    out, saved = _lib.fftconv_fwd(up, kp, bias, save=True)
    du, dk, dbias = _lib.fftconv_bwd(gp, up, kp, bias, saved=saved)
    du2, dk2, dbias2 = _lib.fftconv_bwd(gp, up, kp, bias)                     # the recomputing backward reads u and k again
    assert _lib.ld_of(out) == _lib.ld_of(up) and _lib.ld_of(du) == _lib.ld_of(up) and _lib.ld_of(dk) == _lib.ld_of(kp)
    for a, b in zip(ref, (out, du, dk, dbias)):
        assert torch.equal(a, b)
    for a, b in zip(ref[1:], (du2, dk2, dbias2)):
        assert torch.equal(a, b)
    # nothing is written between the rows: outputs handed in with a sentinel there (through the C ABI directly)
    sent = 12345.0
    outp, outbuf = pitched(torch.zeros_like(u), sent)
    tables = _lib.tables_for(u.device, L)
    ws, stream = _lib.workspace_for(u.device, _lib.lib().hyena_fftconv_workspace_bytes(B, D, L, 1, 0))
    _lib.check(_lib.lib().hyena_fftconv_fwd_ld(up.data_ptr(), kp.data_ptr(), bias.data_ptr(), outp.data_ptr(), B, D, L, ld, ld,
                                               _lib.dtype_code(dtype), tables.data_ptr(), ws.data_ptr(), ws.numel(), 0, None, 0, stream))
    assert torch.equal(outp, ref_out) and (outbuf[..., L:] == sent).all()
    dup, dubuf = pitched(torch.zeros_like(u), sent)
    dkp, dkbuf = pitched(torch.zeros_like(k), sent)
    db = torch.zeros(D)
    _lib.check(_lib.lib().hyena_fftconv_bwd_ld(gp.data_ptr(), up.data_ptr(), kp.data_ptr(), bias.data_ptr(), dup.data_ptr(), dkp.data_ptr(),
                                               db.data_ptr(), B, D, L, ld, ld, _lib.dtype_code(dtype), tables.data_ptr(), ws.data_ptr(),
                                               ws.numel(), 0, None, 0, stream))
    assert torch.equal(dup, ref[1]) and torch.equal(dkp, ref[2]) and (dubuf[..., L:] == sent).all() and (dkbuf[..., L:] == sent).all()
    # a pitch below L is refused
    assert _lib.lib().hyena_fftconv_fwd_ld(up.data_ptr(), kp.data_ptr(), None, outp.data_ptr(), B, D, L, L - 1, ld, _lib.dtype_code(dtype),
                                           tables.data_ptr(), ws.data_ptr(), ws.numel(), 0, None, 0, stream) == 1


@pytest.mark.parametrize("B,D,L,dtype", [(17, 3, 700, torch.bfloat16), (16, 8, 1024, torch.float32), (9, 2, 1025, torch.float16), (8, 8, 2048, torch.bfloat16),
                                         (5, 2, 4096, torch.float32), (3, 2, 8192, torch.bfloat16), (2, 2, 16384, torch.bfloat16), (3, 1, 16383, torch.float32),
                                         (40, 3, 700, torch.bfloat16), (9, 4, 5000, torch.float16)])
def test_du_from_dks_transform_of_dout(emu_backend, monkeypatch, B, D, L, dtype):
    """Round 6: HYENA_FFTCONV_DUDK=1 -- dk_kernel<.., DU = true> also multiplies its transform of dout by conj(H) and inverts it, so
    the separate du launch (which transforms dout a second time) is gone at M <= 16384, B >= 2.  dk and dbias: the same kernel code, the same bits; du:
    conv_kernel's arithmetic (bitwise under the emulator), incl. batch slices, ragged last row groups and pitched rows; the oracle's values."""
    monkeypatch.setenv("HYENA_FFTCONV_SMALL", "0")          # (the one-launch short-row pair would serve the small cases otherwise)
    u, k, bias, dout = _inputs(B, D, L, dtype, seed=B + L)

    def pitched(t):
        r = emu_backend.empty_rows(t.shape[:-1], t.shape[-1], t.dtype, t.device)
        r.copy_(t)
        return r

    res = {}
    for knob in ("0", "1"):
        monkeypatch.setenv("HYENA_FFTCONV_DUDK", knob)
        for lay in ("packed", "pitched"):
            ud, gd, kd = (pitched(u), pitched(dout), pitched(k)) if lay == "pitched" else (u, dout, k)
            out, saved = emu_backend.fftconv_fwd(ud, kd, bias, save=True)
            res[knob, lay, "saved"] = emu_backend.fftconv_bwd(gd, ud, kd, bias, saved=saved)
            res[knob, lay, "recomputed"] = emu_backend.fftconv_bwd(gd, ud, kd, bias)
    ref = res["0", "packed", "saved"]
    for key, (du, dk, dbias) in res.items():
        assert torch.equal(du, ref[0]) and torch.equal(dk, ref[1]) and torch.equal(dbias, ref[2]), key
    r_out, r_du, r_dk, r_db = _oracle(u.float(), k, bias, dout.float())
    tol = 3e-6 if dtype == torch.float32 else (6e-3 if dtype == torch.bfloat16 else 8e-4)
    assert _rel(ref[0].float(), r_du) < tol and _rel(ref[1], r_dk) < 3e-6
