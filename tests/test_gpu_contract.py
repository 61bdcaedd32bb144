"""Operator- and model-level parity AT THE CONTRACT SHAPES on a real MI355X:

* the fused mixer core (channel-major shell + long convolution, ``hyena_mixer_core_cm``) and the whole ``HyenaOperator``
  (fused filter + ``in_proj_cm`` / ``out_proj_cm`` + shell + long conv) against the oracle ``O.hyena_operator`` /
  its pieces at (B, L, D) = (8, 32768, 256), (2, 160000, 256), (1, 2^20, 256) -- forward, input gradient and every
  parameter gradient; in fp32 (tight tolerance: this is the index-map check -- round 1's only hardware bug showed up at
  D = 256 at scale and nowhere else) and under bf16 autocast (the training configuration, 16-bit tolerance);
* ``HyenaDNALM`` against the golden minted from the reference's ``SimpleLMHeadModel`` (oracle/make_golden_lm.py,
  simple_lm.py:26-305): logits, loss and every gradient.

Where the oracle is evaluated.  `oracle/hyena_oracle.py` is pinned on the CPU against the reference-minted goldens
(tests/test_oracle_golden.py).  At the contract shapes the host cores of the GPU box need ~3 minutes per case for it (measured:
seven cases did not fit a 25-minute budget; F.conv1d over 2^20 positions x 768 groups and the fp32 autograd graph dominate), so
here the SAME oracle functions are evaluated in FLOAT64 by PyTorch's own device ops (hipFFT, rocBLAS, element-wise kernels --
none of this package's kernels), with the depthwise convolution in its tap-by-tap form `O.short_conv_taps` (pinned to
`O.short_conv`'s F.conv1d in tests/test_oracle_golden.py); `test_device_evaluated_oracle_equals_host_oracle` ties the two
evaluations together at (2, 8192, 256) in this file, on the GPU box."""
import os

import pytest
import torch

from oracle import hyena_oracle as O

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SHAPES = [(8, 32768, 256), (2, 160000, 256), (1, 1048576, 256),
          # what the reference's trainer really feeds the operator: L = max_length - 1 (hg38_dataset.py:220-223) -- pitched rows, round 5
          (8, 32767, 256), (2, 159999, 256), (1, 1048575, 256)]


def _f64(t):
    return t.detach().to(device=torch.device("cuda", 0), dtype=torch.float64)      # norms of 10^9-element tensors: on the device


def _rel(a, b):
    a, b = _f64(a), _f64(b)
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _per_channel_rel(a, b, dim):
    """worst relative L2 error over the slices along `dim` (a per-channel bound: one bad channel cannot hide in the norm)"""
    a, b = _f64(a), _f64(b)
    dims = [d for d in range(a.dim()) if d != dim]
    num = (a - b).pow(2).sum(dim=dims).sqrt()
    den = b.pow(2).sum(dim=dims).sqrt().clamp_min(1e-30)
    return (num / den).max().item()


def _ref_core_cm(xT, b_in, w, b, k, bias, L, short_conv=O.short_conv_taps):
    D = xT.shape[0] // 3
    x = (xT + b_in[:, None, None]).permute(1, 0, 2)
    xc = short_conv(x, w, b, L)
    x0, x1, v = xc.split(D, dim=1)
    return (O.fftconv_ref(v * x1, k, bias) * x0).permute(1, 0, 2)


def test_device_evaluated_oracle_equals_host_oracle(gpu_lib):
    """the oracle functions evaluated by PyTorch's device ops in float64 (what the contract-shape cases below compare with) against
    their evaluation on the host in fp32 with F.conv1d (what tests/test_oracle_golden.py pins to the reference) -- same function,
    two executors, at a size the host does in seconds"""
    dev = torch.device("cuda", 0)
    B, L, D = 2, 8192, 256
    op, u, dy, ref = _operator_and_oracle(B, L, D, seed=77, device=torch.device("cpu"), dtype=torch.float32, short_conv=O.short_conv)
    _, _, _, ref64 = _operator_and_oracle(B, L, D, seed=77, device=dev, dtype=torch.float64, short_conv=O.short_conv_taps)
    assert _rel(ref["y"], ref64["y"]) < 5e-6 and _rel(ref["du"], ref64["du"]) < 2e-5
    for n, g in ref["grads"].items():
        if g is not None:
            assert _rel(g, ref64["grads"][n]) < 5e-5, n


@pytest.mark.parametrize("B,L,D", SHAPES)
def test_mixer_core_cm_at_contract_shapes_vs_oracle(gpu_lib, B, L, D):
    """hyena.py:392-439 between the projections, bf16 tensors as under autocast; oracle in float64 on the same bf16 inputs"""
    from hyena_dna_amd.mixer import hyena_mixer_core_cm
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(L + D)
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)      # noqa: E731
    xT = rn(3 * D, B, L).to(torch.bfloat16)
    b_in = rn(3 * D) * 0.3
    w = rn(3 * D, 1, 3) * 0.5
    b = rn(3 * D) * 0.2
    k = rn(D, L) * torch.exp(-5.0 * torch.linspace(0, 1, L, device=dev))[None] * 0.1
    bias = rn(D)
    dz = rn(D, B, L).to(torch.bfloat16)
    leaves = [t.clone().requires_grad_(True) for t in (xT, b_in, w, b, k, bias)]
    z = hyena_mixer_core_cm(*leaves, L)
    z.backward(dz)
    ref_leaves = [t.detach().double().requires_grad_(True) for t in (xT, b_in, w, b, k, bias)]      # float64, PyTorch device ops
    zr = _ref_core_cm(*ref_leaves, L)
    zr.backward(dz.double())
    del zr
    torch.cuda.empty_cache()
    # 16-bit storage of vg, y, z and of the gradients between the kernels: ~3 roundings of 2^-9 on each path
    zr = _ref_core_cm(*[t.detach() for t in ref_leaves], L)
    assert _rel(z.float(), zr) < 8e-3 and _per_channel_rel(z.float(), zr, 0) < 1.2e-2
    del zr
    assert _rel(leaves[0].grad.float(), ref_leaves[0].grad) < 1.2e-2
    assert _per_channel_rel(leaves[0].grad.float(), ref_leaves[0].grad, 0) < 2e-2
    for n, a, r in zip(["db_in", "dw_sc", "db_sc", "dk", "dbias"], leaves[1:], ref_leaves[1:]):
        e = _rel(a.grad.float(), r.grad)
        assert e < 1.2e-2, (n, e)
    assert _per_channel_rel(leaves[4].grad.float(), ref_leaves[4].grad, 0) < 2e-2


def _operator_and_oracle(B, L, D, seed, device=None, dtype=torch.float64, short_conv=O.short_conv_taps, filter_fn=None, order=2):
    """(operator, u, dy, oracle results): the oracle's HyenaOperator.forward (O.hyena_operator) + autograd, evaluated on `device`
    (default cuda:0) in `dtype`"""
    from hyena_dna_amd.hyena import HyenaOperator
    device = device or torch.device("cuda", 0)
    torch.manual_seed(seed)
    op = HyenaOperator(d_model=D, l_max=L + 2, order=order, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10,
                       lr=6e-4, wd=0.0, lr_pos_emb=0.0)
    with torch.no_grad():                       # biases are zero-initialised by the LM; give them values here
        for n, p in op.named_parameters():
            if n.endswith("bias") and n != "filter_fn.bias":
                p.normal_(0, 0.1)
    sd = {k_: v.detach().clone() for k_, v in op.state_dict().items()}
    g = torch.Generator().manual_seed(seed + 1)
    u = torch.randn(B, L, D, generator=g)
    dy = torch.randn(B, L, D, generator=g)
    # oracle: the reference's forward restated (hyena.py:388-444), fp32 on the host, autograd for the gradients
    params = dict(op.named_parameters())
    leaves = {k_: v.to(device=device, dtype=dtype if v.is_floating_point() else v.dtype).requires_grad_(v.is_floating_point() and k_ in params)
              for k_, v in sd.items()}
    for i in (3, 5):                            # hyena.py:199: ONE freq parameter shared by the three activations
        leaves[f"filter_fn.implicit_filter.{i}.freq"] = leaves["filter_fn.implicit_filter.1.freq"]
    u_ref = u.to(device=device, dtype=dtype).requires_grad_(True)
    y_ref = O.hyena_operator(leaves, u_ref, l_max=L + 2, order=order, short_conv_fn=short_conv, filter_fn=filter_fn)
    y_ref.backward(dy.to(device=device, dtype=dtype))
    ref = dict(y=y_ref.detach().cpu(), du=u_ref.grad.cpu(), grads={n: None if leaves[n].grad is None else leaves[n].grad.cpu() for n in params})
    del leaves, u_ref, y_ref
    if device.type == "cuda":
        torch.cuda.empty_cache()
    return op, u, dy, ref


@pytest.mark.parametrize("B,L,D", SHAPES)
def test_operator_at_contract_shapes_vs_oracle(gpu_lib, B, L, D, monkeypatch):
    """The whole HyenaOperator.forward (hyena.py:388-444) on the fused path vs the oracle: fp32; bf16 autocast with the fp32 filter
    kernels (HYENA_FILTER_AUTOCAST=fp32: everything but the filter at 16-bit tolerance of the fp64 oracle); bf16 autocast as it runs
    by default -- the filter from the 16-bit kernels -- against the oracle with the filter evaluated as the reference evaluates it
    under autocast (O.hyena_filter_autocast, PyTorch's device GEMMs): that graph's 16-bit roundings sit in front of sin(10 a), so
    rounding flips between two fp32 summation orders move ~5 % of the positions by a few % (tests/test_gpu_filter.py) -- looser bounds."""
    dev = torch.device("cuda", 0)
    op, u, dy, ref = _operator_and_oracle(B, L, D, seed=L // 7 + B)
    ref16 = _operator_and_oracle(B, L, D, seed=L // 7 + B, filter_fn=O.hyena_filter_autocast)[3]
    op = op.to(dev)
    assert op._fused_ok()
    for mode in ("fp32", "bf16", "bf16-filter16"):
        op.zero_grad(set_to_none=True)
        ud = u.to(dev).requires_grad_(True)
        if mode == "bf16-filter16":
            ref = ref16
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = op(ud.to(torch.bfloat16))
            y.float().backward(dy.to(dev))
            ty, tg = 3e-2, 6e-2
        elif mode == "fp32":
            y = op(ud)
            y.backward(dy.to(dev))
            # fp32 everywhere (library GEMMs in fp32, fused filter, exact-fp32 transforms): rounding noise only.  The
            # filter's sin(10 x) chain amplifies one fp32 rounding to ~1e-6, the GEMMs sum 256-768 terms.
            ty, tg = 2e-5, 2e-4
        else:
            monkeypatch.setenv("HYENA_FILTER_AUTOCAST", "fp32")
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = op(ud.to(torch.bfloat16))
            y.float().backward(dy.to(dev))
            monkeypatch.delenv("HYENA_FILTER_AUTOCAST")
            ty, tg = 1.5e-2, 3e-2
        assert y.shape == ref["y"].shape
        e = _rel(y.float(), ref["y"])
        assert e < ty, (mode, "y", e)
        assert _per_channel_rel(y.float(), ref["y"], 2) < 3 * ty, (mode, "y per channel")
        e = _rel(ud.grad.float(), ref["du"])
        assert e < tg, (mode, "du", e)
        for n, p in op.named_parameters():
            r = ref["grads"][n]
            if r is None:
                assert p.grad is None or torch.count_nonzero(p.grad) == 0, n
                continue
            e = _rel(p.grad.float(), r)
            assert e < tg, (mode, n, e)
        del y, ud
        torch.cuda.empty_cache()


ORDER_SHAPES = [(4, 1023, 128, 3), (2, 8191, 128, 3), (1, 131071, 256, 3), (2, 32767, 256, 4)]


@pytest.mark.parametrize("B,L,D,order", ORDER_SHAPES)
def test_operator_of_higher_order_vs_oracle(gpu_lib, B, L, D, order, monkeypatch):
    """order 3 / 4 (configs/model/layer/hyena_dna.yaml:3; hyena.py:404-439) on the channel-major route of round 6 (mixer.HyenaMixerCMOrderNFunc)
    at the trainer's odd lengths, against O.hyena_operator in float64: fp32, and bf16 autocast with the fp32 filter kernels; then the generic
    route (the reference's graph around the HIP convolution) against the same oracle values in fp32"""
    import hyena_dna_amd.hyena as H
    dev = torch.device("cuda", 0)
    op, u, dy, ref = _operator_and_oracle(B, L, D, seed=L // 5 + order, order=order)
    op = op.to(dev)
    for mode in ("fp32", "bf16", "generic-fp32"):
        monkeypatch.setattr(H, "ORDER_N_FUSED", mode != "generic-fp32")
        assert op._route(L) == ("generic" if mode == "generic-fp32" else "order_n")
        op.zero_grad(set_to_none=True)
        ud = u.to(dev).requires_grad_(True)
        if mode == "bf16":
            monkeypatch.setenv("HYENA_FILTER_AUTOCAST", "fp32")
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = op(ud.to(torch.bfloat16))
            y.float().backward(dy.to(dev))
            monkeypatch.delenv("HYENA_FILTER_AUTOCAST")
            ty, tg = 2e-2, 4e-2                   # (one more 16-bit gate and convolution per extra order than the order-2 bounds above)
        else:
            y = op(ud)
            y.backward(dy.to(dev))
            ty, tg = 3e-5, 3e-4
        e = _rel(y.float(), ref["y"])
        assert e < ty, (mode, "y", e)
        assert _per_channel_rel(y.float(), ref["y"], 2) < 3 * ty, (mode, "y per channel")
        e = _rel(ud.grad.float(), ref["du"])
        assert e < tg, (mode, "du", e)
        for n, p in op.named_parameters():
            r = ref["grads"][n]
            if r is None:
                assert p.grad is None or torch.count_nonzero(p.grad) == 0, n
                continue
            e = _rel(p.grad.float(), r)
            assert e < tg, (mode, n, e)
        del y, ud
        torch.cuda.empty_cache()


# The two round-5 matrix-core kernels behind user-reachable knobs, held to the ORACLE (not to this package's other kernels: VERDICT r5 weak 1) at
# the reference trainer's own shapes.  B < 8 is where the auto policy does not take the dgrad kernel and the add + LayerNorm epilogue is off by default.
KNOB_SHAPES = [(2, 159999, 256), (1, 1048575, 256), (8, 32767, 256)]


@pytest.mark.parametrize("B,L,D", KNOB_SHAPES)
def test_operator_with_outproj_dgrad_kernel_forced_vs_oracle(gpu_lib, B, L, D, monkeypatch):
    """HYENA_OUTPROJ_DGRAD_MFMA=1: out_proj's input gradient + the gate's backward in one matrix-core kernel (outproj_dgrad_gate_bwd_kernel),
    at every B -- the whole operator, bf16 autocast, against O.hyena_operator in float64: output, input gradient, every parameter gradient"""
    import hyena_dna_amd.mixer as mixer
    from hyena_dna_amd import _lib
    dev = torch.device("cuda", 0)
    monkeypatch.setattr(mixer, "DGRAD_MFMA", True)
    monkeypatch.setenv("HYENA_FILTER_AUTOCAST", "fp32")          # everything around the filter at the tight 16-bit bounds (as the "bf16" mode above)
    assert _lib.outproj_dgrad_supported(B, L, D, torch.bfloat16) and mixer._dgrad_fused(B, L, D, torch.bfloat16)
    op, u, dy, ref = _operator_and_oracle(B, L, D, seed=L // 5 + B)
    op = op.to(dev)
    calls = []
    real = _lib.outproj_dgrad_gate_bwd
    monkeypatch.setattr(_lib, "outproj_dgrad_gate_bwd", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    ud = u.to(dev).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = op(ud.to(torch.bfloat16))
    y.float().backward(dy.to(dev))
    assert calls, "the forced knob did not route the backward through the dgrad kernel"
    ty, tg = 1.5e-2, 3e-2
    assert _rel(y.float(), ref["y"]) < ty and _per_channel_rel(y.float(), ref["y"], 2) < 3 * ty
    e = _rel(ud.grad.float(), ref["du"])
    assert e < tg, ("du", e)
    assert _per_channel_rel(ud.grad.float(), ref["du"], 2) < 3 * tg
    for n, p in op.named_parameters():
        r = ref["grads"][n]
        if r is None:
            assert p.grad is None or torch.count_nonzero(p.grad) == 0, n
            continue
        e = _rel(p.grad.float(), r)
        assert e < tg, (n, e)


@pytest.mark.parametrize("B,L,D", KNOB_SHAPES)
def test_operator_with_add_norm_epilogue_forced_vs_oracle(gpu_lib, B, L, D, monkeypatch):
    """HYENA_ADD_NORM_FUSED=1: out_proj + the block's residual add + LayerNorm in one kernel (HyenaOperator.forward_add_norm) against the oracle's
    operator followed by an fp64 add + LayerNorm (simple_lm.py:280-284): hidden, residual', and -- for upstream gradients on BOTH outputs -- the
    gradients of u, the incoming residual, the norm's weight / bias and every parameter of the operator"""
    import hyena_dna_amd.hyena as hy
    dev = torch.device("cuda", 0)
    monkeypatch.setattr(hy, "ADD_NORM_FUSED", True)
    monkeypatch.setenv("HYENA_FILTER_AUTOCAST", "fp32")
    seed = L // 3 + B
    g = torch.Generator().manual_seed(seed + 9)
    res = torch.randn(B, L, D, generator=g)
    ln_w = 1.0 + 0.2 * torch.randn(D, generator=g)
    ln_b = 0.1 * torch.randn(D, generator=g)
    dh = torch.randn(B, L, D, generator=g)
    dr = 0.5 * torch.randn(B, L, D, generator=g)
    eps = 1e-5

    # ---- oracle: O.hyena_operator -> + residual -> LayerNorm, float64 on the device's PyTorch ops ----
    from hyena_dna_amd.hyena import HyenaOperator
    torch.manual_seed(seed)
    op = HyenaOperator(d_model=D, l_max=L + 2, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10,
                       lr=6e-4, wd=0.0, lr_pos_emb=0.0)
    with torch.no_grad():
        for n, p in op.named_parameters():
            if n.endswith("bias") and n != "filter_fn.bias":
                p.normal_(0, 0.1)
    u = torch.randn(B, L, D, generator=g)
    params = dict(op.named_parameters())
    leaves = {k_: v.detach().to(device=dev, dtype=torch.float64 if v.is_floating_point() else v.dtype).requires_grad_(v.is_floating_point() and k_ in params)
              for k_, v in op.state_dict().items()}
    for i in (3, 5):
        leaves[f"filter_fn.implicit_filter.{i}.freq"] = leaves["filter_fn.implicit_filter.1.freq"]
    u64 = u.to(dev, torch.float64).requires_grad_(True)
    r64 = res.to(dev, torch.float64).requires_grad_(True)
    w64 = ln_w.to(dev, torch.float64).requires_grad_(True)
    b64 = ln_b.to(dev, torch.float64).requires_grad_(True)
    y64 = O.hyena_operator(leaves, u64, l_max=L + 2, short_conv_fn=O.short_conv_taps)
    rp64 = y64 + r64
    h64 = torch.nn.functional.layer_norm(rp64, (D,), w64, b64, eps)
    torch.autograd.backward([h64, rp64], [dh.to(dev, torch.float64), dr.to(dev, torch.float64)])
    ref = dict(h=h64.detach().cpu(), rp=rp64.detach().cpu(), du=u64.grad.cpu(), dres=r64.grad.cpu(), dw=w64.grad.cpu(), db=b64.grad.cpu(),
               grads={n: None if leaves[n].grad is None else leaves[n].grad.cpu() for n in params})
    del leaves, u64, r64, y64, rp64, h64
    torch.cuda.empty_cache()

    # ---- this package: forward_add_norm under bf16 autocast ----
    op = op.to(dev)
    ud = u.to(dev).requires_grad_(True)
    rd = res.to(dev).requires_grad_(True)                         # the fp32 residual stream (residual_in_fp32)
    wd = ln_w.to(dev).requires_grad_(True)
    bd = ln_b.to(dev).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = op.forward_add_norm(ud.to(torch.bfloat16), rd, wd, bd, eps)
    assert out is not None, "the forced knob did not take the fused add + LayerNorm route"
    h, rp = out
    assert rp.dtype == torch.float32 and h.shape == (B, L, D) and rp.shape == (B, L, D)
    torch.autograd.backward([h.float(), rp], [dh.to(dev), dr.to(dev)])
    ty, tg = 1.5e-2, 3e-2
    assert _rel(h.float(), ref["h"]) < ty and _rel(rp, ref["rp"]) < ty
    assert _per_channel_rel(h.float(), ref["h"], 2) < 3 * ty
    for name, a, r in (("du", ud.grad, ref["du"]), ("dresidual", rd.grad, ref["dres"]), ("d ln weight", wd.grad, ref["dw"]),
                       ("d ln bias", bd.grad, ref["db"])):
        e = _rel(a.float(), r)
        assert e < tg, (name, e)
    for n, p in op.named_parameters():
        r = ref["grads"][n]
        if r is None:
            assert p.grad is None or torch.count_nonzero(p.grad) == 0, n
            continue
        e = _rel(p.grad.float(), r)
        assert e < tg, (n, e)


def test_lm_vs_reference_simple_lm_golden(gpu_lib):
    """HyenaDNALM (2 layers, d_model 128, L = 4096) vs SimpleLMHeadModel's logits, loss and every gradient (the golden is the
    reference's own model on the CPU in fp32)."""
    from hyena_dna_amd.lm import HyenaDNALM
    dev = torch.device("cuda", 0)
    c = torch.load(os.path.join(GOLDEN, "lm_simple_d128_l4096.pt"), weights_only=False)
    for fused_ln in (True, False):
        model = HyenaDNALM(layer=dict(c["layer"]), fused_dropout_add_ln=fused_ln, **c["cfg"])
        model.load_state_dict(c["state_dict"], strict=True)
        model = model.to(dev)
        ids, tgt = c["ids"].to(dev), c["targets"].to(dev)
        logits = model(ids)[0].logits
        loss = torch.nn.functional.cross_entropy(logits.float().reshape(-1, logits.shape[-1]), tgt.reshape(-1))
        loss.backward()
        assert _rel(logits, c["logits"]) < 2e-5
        assert abs(loss.item() - c["loss"]) < 1e-5 * abs(c["loss"]) + 1e-6
        grads = {n: p.grad for n, p in model.named_parameters()}
        assert set(grads) == set(c["grads"])
        for n, gref in c["grads"].items():
            e = _rel(grads[n], gref)
            assert e < 5e-4, (fused_ln, n, e)
    # the training configuration: bf16 autocast, same weights -- 16-bit tolerance on the logits and the loss
    model.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        logits = model(ids)[0].logits
    loss = torch.nn.functional.cross_entropy(logits.float().reshape(-1, logits.shape[-1]), tgt.reshape(-1))
    loss.backward()
    # (the filter now follows the reference's autocast graph -- 16-bit Linear layers -- where the golden is the fp32 model: at the LM's
    # initialisation, N(0, 0.02) weights, that graph is ~5e-3 from the fp32 filter)
    assert _rel(logits.float(), c["logits"]) < 5e-2 and abs(loss.item() - c["loss"]) < 2e-2 * abs(c["loss"])
    for n, p in model.named_parameters():
        e = _rel(p.grad.float(), c["grads"][n])
        assert e < 0.1, (n, e)
