"""Fused mixer shell (short conv + gates + layout changes) on the CPU-emulated kernels vs the oracle."""
import pytest
import torch
import torch.nn.functional as F

from oracle import hyena_oracle as O


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def _ref_core(x, w, b, k, bias, L):
    """oracle pieces: short conv (hyena.py:363-369,394) -> split -> gates -> fftconv_ref -> transpose"""
    D = x.shape[-1] // 3
    xc = O.short_conv(x.transpose(1, 2), w, b, L)
    x0, x1, v = xc.split(D, dim=1)
    y = O.fftconv_ref(v * x1, k, bias)
    return (y * x0).transpose(1, 2)


@pytest.mark.parametrize("B,Lx,L,D", [(2, 70, 70, 8), (1, 1500, 1500, 5), (2, 2100, 2048, 70), (1, 130, 64, 64),
                                      (2, 1023, 1023, 128), (1, 2500, 2049, 64), (1, 66, 66, 192), (2, 7, 7, 64)])
def test_mixer_core_fp32_vs_oracle(emu_backend, B, Lx, L, D):
    from hyena_dna_amd.mixer import hyena_mixer_core
    g = torch.Generator().manual_seed(B * 1000 + L + D)
    x = torch.randn(B, Lx, 3 * D, generator=g)
    w = torch.randn(3 * D, 1, 3, generator=g) * 0.5
    b = torch.randn(3 * D, generator=g) * 0.1
    k = torch.randn(D, L, generator=g) * torch.exp(-4 * torch.linspace(0, 1, L)) * 0.2
    bias = torch.randn(D, generator=g)
    dz = torch.randn(B, L, D, generator=g)

    def run(fn):
        ts = [t.clone().requires_grad_(True) for t in (x, w, b, k, bias)]
        z = fn(*ts)
        z.backward(dz)
        return [z.detach()] + [t.grad for t in ts]

    got = run(lambda x_, w_, b_, k_, bias_: hyena_mixer_core(x_, w_, b_, k_, bias_, L))
    ref = run(lambda x_, w_, b_, k_, bias_: _ref_core(x_, w_, b_, k_, bias_, L))
    names = ["z", "dx", "dw", "db", "dk", "dbias"]
    for n, a, r in zip(names, got, ref):
        assert a.shape == r.shape, n
        assert _rel(a, r) < 5e-6, (n, _rel(a, r))
    if Lx > L:
        assert torch.count_nonzero(got[1][:, L:]) == 0        # truncated positions get no gradient


def test_mixer_core_bf16(emu_backend):
    from hyena_dna_amd.mixer import hyena_mixer_core
    g = torch.Generator().manual_seed(3)
    B, L, D = 2, 301, 64
    x = torch.randn(B, L, 3 * D, generator=g).bfloat16()
    w = torch.randn(3 * D, 1, 3, generator=g) * 0.5
    b = torch.randn(3 * D, generator=g) * 0.1
    k = torch.randn(D, L, generator=g) * torch.exp(-4 * torch.linspace(0, 1, L)) * 0.2
    bias = torch.randn(D, generator=g)
    z = hyena_mixer_core(x, w, b, k, bias, L)
    assert z.dtype == torch.bfloat16
    ref = _ref_core(x.float(), w, b, k, bias, L)
    assert _rel(z.float(), ref) < 6e-3                        # two bf16 roundings (vg, z) on an fp32 pipeline


@pytest.mark.parametrize("name", ["d8l64", "d16l257", "d8l80_trunc"])
def test_operator_fused_path_matches_reference_vectors(emu_backend, golden_operator, name):
    """HyenaOperator mirror takes the fused path for the HyenaDNA configuration: same outputs and gradients as the
    reference module (golden vectors minted from it)."""
    from hyena_dna_amd.hyena import HyenaOperator
    c = golden_operator[name]
    op = HyenaOperator(d_model=c["d_model"], l_max=c["l_max"], order=2, filter_order=64, emb_dim=5,
                       short_filter_order=3, modulate=True, w=10, lr=6e-4, wd=0.0, lr_pos_emb=0.0)
    op.load_state_dict(c["state_dict"])
    assert op._fused_ok()
    u = c["u"].clone().requires_grad_(True)
    y = op(u)
    torch.testing.assert_close(y, c["y"], rtol=1e-5, atol=2e-6)
    y.backward(c["dy"])
    torch.testing.assert_close(u.grad, c["du"], rtol=1e-4, atol=2e-6)
    for n, p in op.named_parameters():
        torch.testing.assert_close(p.grad, c["grads"][n], rtol=2e-4, atol=2e-5, msg=lambda m, n=n: f"{n}: {m}")
    # and the generic path (order 3 is not covered by the fused core) still runs
    op3 = HyenaOperator(d_model=8, l_max=40, order=3, filter_order=16, emb_dim=5)
    assert not op3._fused_ok()
    assert op3(torch.randn(1, 40, 8)).shape == (1, 40, 8)
