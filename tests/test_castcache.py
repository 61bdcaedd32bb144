"""hyena_dna_amd/_castcache.py (round 6): per-step 16-bit shadows of the fp32 parameters, refreshed in one batched copy."""
import torch

from hyena_dna_amd import _castcache as CC


def test_shadows_follow_the_parameters_and_refresh_in_bulk(monkeypatch):
    monkeypatch.setattr(CC, "ENABLED", True)
    CC.reset()
    ps = [torch.nn.Parameter(torch.randn(7, 5)), torch.nn.Parameter(torch.randn(9)), torch.nn.Parameter(torch.randn(3, 4))]
    for p in ps:                                    # first uses: registered one by one
        assert torch.equal(CC.shadow(p, torch.bfloat16), p.detach().to(torch.bfloat16))
    b0 = CC.rounded_f32(ps[1], torch.bfloat16)
    assert b0.dtype == torch.float32 and torch.equal(b0, ps[1].detach().to(torch.bfloat16).float())
    n0 = CC.stats()["bulk_refreshes"]
    s0 = CC.shadow(ps[0], torch.bfloat16)
    assert CC.shadow(ps[0], torch.bfloat16) is s0 and CC.stats()["bulk_refreshes"] == n0          # fresh: a hit, the same tensor
    with torch.no_grad():                           # an "optimizer step": every parameter moves
        for p in ps:
            p.add_(1.0)
    s1 = CC.shadow(ps[2], torch.bfloat16)           # the first use after it refreshes ALL of them in one pass
    assert CC.stats()["bulk_refreshes"] == n0 + 1
    for p in ps:
        assert torch.equal(CC.shadow(p, torch.bfloat16), p.detach().to(torch.bfloat16))
    assert torch.equal(CC.rounded_f32(ps[1], torch.bfloat16), ps[1].detach().to(torch.bfloat16).float())
    assert CC.stats()["bulk_refreshes"] == n0 + 1 and s1 is CC.shadow(ps[2], torch.bfloat16)
    # a second compute type has its own shadows; plain tensors / 16-bit parameters / fp32 targets are cast the ordinary way
    assert torch.equal(CC.shadow(ps[0], torch.float16), ps[0].detach().to(torch.float16))
    t = torch.randn(4, requires_grad=True)
    assert torch.equal(CC.shadow(t, torch.bfloat16), t.detach().to(torch.bfloat16)) and (id(t), torch.bfloat16) not in CC._entries
    CC.invalidate()
    CC.shadow(ps[0], torch.bfloat16)
    assert CC.stats()["bulk_refreshes"] >= n0 + 2
    del ps, p
    import gc
    gc.collect()
    CC.shadow(torch.nn.Parameter(torch.randn(2, 2)), torch.bfloat16)      # a refresh drops the entries of dead parameters
    assert CC.stats()["entries"] <= 2


def test_linear_under_autocast_semantics_with_and_without_the_cache(monkeypatch):
    """projection.SplitKLinearFunc with the fp32 parameters + compute type (the shadow route) against the plain casts: same output bits; the weight
    gradient in the parameter's type -- fp32 as accumulated by default, through the 16-bit type with ROUND_WGRAD (autocast's own graph)"""
    from hyena_dna_amd.projection import SplitKLinearFunc
    torch.manual_seed(0)
    w, b = torch.nn.Parameter(torch.randn(6, 8) * 0.3), torch.nn.Parameter(torch.randn(6) * 0.1)
    x = torch.randn(40, 8).to(torch.bfloat16).requires_grad_(True)
    dy = torch.randn(40, 6).to(torch.bfloat16)
    ref = torch.nn.functional.linear(x, w.to(torch.bfloat16), b.to(torch.bfloat16))
    ref.backward(dy)
    gref = (x.grad.clone(), w.grad.clone(), b.grad.clone())
    for rnd in (False, True):
        monkeypatch.setattr(CC, "ROUND_WGRAD", rnd)
        x.grad = w.grad = b.grad = None
        y = SplitKLinearFunc.apply(x, w, b, torch.bfloat16)
        assert torch.equal(y, ref)
        y.backward(dy)
        assert w.grad.dtype == torch.float32 and b.grad.dtype == torch.float32 and torch.equal(x.grad, gref[0])
        exact_w = dy.float().t() @ x.detach().float()
        if rnd:
            assert torch.equal(w.grad, exact_w.to(torch.bfloat16).float())
        else:
            assert torch.allclose(w.grad, exact_w, rtol=1e-5, atol=1e-5)
        assert torch.allclose(w.grad, gref[1], rtol=2e-2, atol=2e-2) and torch.allclose(b.grad, gref[2], rtol=2e-2, atol=5e-2)
