"""Worker of tests/test_overlay_reference.py: runs the UNMODIFIED reference module src.models.sequence.hyena (from
/root/reference) with this repository's path overlay in front of it, kernels under tests/hipemu.  Build container only."""
import importlib
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("HYENA_REFERENCE", "/root/reference")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def main():
    # overlay first, then this repo, then the reference
    sys.path[:0] = [os.path.join(ROOT, "overlay"), ROOT, REF]

    def _get(path):
        mod, _, attr = path.rpartition(".")
        return getattr(importlib.import_module(mod), attr)

    # inert stand-ins for packages that are not installed here; they touch no arithmetic
    _stub("hydra", utils=_stub("hydra.utils", get_method=_get, get_class=_get))
    _stub("omegaconf", ListConfig=list, DictConfig=type("DictConfig", (dict,), {}), OmegaConf=object)
    _stub("pytorch_lightning", utilities=_stub("pytorch_lightning.utilities", rank_zero_only=lambda f: f))
    _stub("opt_einsum", contract=torch.einsum)

    from hyena_dna_amd import _lib
    from tests.hipemu.emu_backend import EmuBackend
    _lib._backend = EmuBackend()

    import src.ops.fftconv as ops_fftconv                       # must be the overlay's file, not the reference's
    assert os.path.realpath(ops_fftconv.__file__).startswith(os.path.realpath(os.path.join(ROOT, "overlay"))), ops_fftconv.__file__
    import src.models.sequence.hyena as ref_hyena               # the reference's own, unmodified module
    assert os.path.realpath(ref_hyena.__file__).startswith(os.path.realpath(REF)), ref_hyena.__file__
    assert ref_hyena.fftconv_func is ops_fftconv.fftconv_func and ref_hyena.fftconv_func is not None

    torch.manual_seed(0)
    kw = dict(d_model=16, l_max=130, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10,
              lr=6e-4, wd=0.0, lr_pos_emb=0.0)
    fused = ref_hyena.HyenaOperator(fused_fft_conv=True, **kw)      # -> fftconv_func (hyena.py:250-259) = the HIP op
    plain = ref_hyena.HyenaOperator(fused_fft_conv=False, **kw)     # -> the reference's torch.fft path (hyena.py:261)
    plain.load_state_dict(fused.state_dict())
    u = torch.randn(2, 128, 16)
    dy = torch.randn(2, 128, 16)
    outs = []
    for op in (fused, plain):
        x = u.clone().requires_grad_(True)
        y = op(x)
        y.backward(dy)
        outs.append([y.detach(), x.grad] + [p.grad for _, p in sorted(op.named_parameters())])
    worst = 0.0
    for a, b in zip(*outs):
        err = ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
        worst = max(worst, err)
        assert err < 1e-5, err
    print(f"OVERLAY_OK tensors={len(outs[0])} worst_rel={worst:.2e}", flush=True)

    # the registry seam -- the reference's own name-based instantiate builds THIS package's
    # operator (fused mixer shell + fused filter) and a reference checkpoint loads into it
    import src.utils.registry as registry
    from src.utils.config import instantiate
    registry.layer["hyena"] = "hyena_dna_amd.hyena.HyenaOperator"
    registry.layer["hyena-filter"] = "hyena_dna_amd.hyena.HyenaFilter"
    cfg = dict(_name_="hyena", l_max=130, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10,
               lr=6e-4, wd=0.0, lr_pos_emb=0.0)
    kw64 = dict(kw, d_model=64)
    mine = instantiate(registry.layer, cfg, 64)                                  # as create_mixer_cls does (long_conv_lm.py:88-95)
    assert type(mine).__module__ == "hyena_dna_amd.hyena" and cfg["_name_"] == "hyena"
    ref64 = ref_hyena.HyenaOperator(fused_fft_conv=False, **kw64)
    mine.load_state_dict(ref64.state_dict(), strict=True)
    u64, dy64 = torch.randn(2, 128, 64), torch.randn(2, 128, 64)
    res = []
    for op in (mine, ref64):
        x = u64.clone().requires_grad_(True)
        y = op(x)
        y.backward(dy64)
        res.append([y.detach(), x.grad] + [p.grad for _, p in sorted(op.named_parameters())])
    worst2 = 0.0
    for a, b in zip(*res):
        err = ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
        worst2 = max(worst2, err)
        assert err < 2e-5, err
    print(f"REGISTRY_OK tensors={len(res[0])} worst_rel={worst2:.2e}", flush=True)


if __name__ == "__main__":
    main()
