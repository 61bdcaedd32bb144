"""Worker of tests/test_overlay_reference.py::test_reference_backbone_builds_on_the_flash_attn_surface: the UNMODIFIED
reference ``ConvLMHeadModel`` (src/models/sequence/long_conv_lm.py) imported with this repository's ``overlay/`` in front
(``flash_attn`` import surface + ``src.ops.fftconv``), its mixer swapped through the reference's own registry, against
(a) the reference's standalone PyTorch restatement ``SimpleLMHeadModel`` (simple_lm.py) with the same weights and
(b) this package's Lightning-free ``HyenaDNALM``.  Kernels under tests/hipemu.  Build container only."""
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
    sys.path[:0] = [os.path.join(ROOT, "overlay"), ROOT, REF]

    def _get(path):
        mod, _, attr = path.rpartition(".")
        return getattr(importlib.import_module(mod), attr)

    _stub("hydra", utils=_stub("hydra.utils", get_method=_get, get_class=_get))
    _stub("omegaconf", ListConfig=list, DictConfig=type("DictConfig", (dict,), {}), OmegaConf=object)
    _stub("pytorch_lightning", utilities=_stub("pytorch_lightning.utilities", rank_zero_only=lambda f: f))
    _stub("opt_einsum", contract=torch.einsum)
    import transformers.tokenization_utils  # noqa: F401  (probes torchvision; must come before the stub below)

    class _SD(torch.nn.Module):                       # torchvision.ops.StochasticDepth with p = 0: identity
        def __init__(self, p, mode):
            super().__init__()

        def forward(self, x):
            return x
    _stub("torchvision", ops=_stub("torchvision.ops", StochasticDepth=_SD))

    from hyena_dna_amd import _lib
    from tests.hipemu.emu_backend import EmuBackend
    _lib._backend = EmuBackend()

    import flash_attn
    assert os.path.realpath(flash_attn.__file__).startswith(os.path.realpath(os.path.join(ROOT, "overlay")))
    import src.models.sequence.long_conv_lm as ref_lm           # the reference's own, unmodified module
    assert os.path.realpath(ref_lm.__file__).startswith(os.path.realpath(REF)), ref_lm.__file__
    import src.models.sequence.simple_lm as ref_simple
    import src.utils.registry as registry
    assert ref_lm.dropout_add_layer_norm is not None            # the fused add + LayerNorm is importable

    D, L, NL, V = 64, 130, 2, 12
    layer = dict(_name_="hyena", l_max=L + 2, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10,
                 lr=6e-4, wd=0.0, lr_pos_emb=0.0)
    common = dict(d_model=D, n_layer=NL, d_inner=4 * D, vocab_size=V, resid_dropout=0.0, embed_dropout=0.0,
                  pad_vocab_size_multiple=8, residual_in_fp32=True)

    # (b) the reference model, reference mixer (pure PyTorch) -- the baseline
    torch.manual_seed(0)
    simple = ref_simple.SimpleLMHeadModel(layer=dict(layer), **common)
    # (a) the reference ConvLMHeadModel on this repo's flash_attn surface, with this repo's mixer via the registry
    registry.layer["hyena"] = "hyena_dna_amd.hyena.HyenaOperator"
    registry.layer["hyena-filter"] = "hyena_dna_amd.hyena.HyenaFilter"
    conv = ref_lm.ConvLMHeadModel(layer=dict(layer), fused_dropout_add_ln=True, **common)
    assert type(conv.backbone.layers[0]).__module__ == "hyena_dna_amd.lm"
    assert type(conv.backbone.layers[0].mixer).__module__ == "hyena_dna_amd.hyena"
    missing, unexpected = conv.load_state_dict(simple.state_dict(), strict=True)
    # (c) this package's Lightning-free model
    from hyena_dna_amd.lm import HyenaDNALM
    mine = HyenaDNALM(layer=dict(layer), fused_dropout_add_ln=True, **common)
    mine.load_state_dict(simple.state_dict(), strict=True)

    ids = torch.randint(7, 11, (2, L))
    tgt = torch.roll(ids, -1, 1)
    outs = []
    for m in (simple, conv, mine):
        m.zero_grad(set_to_none=True)
        logits = m(ids)[0].logits
        loss = torch.nn.functional.cross_entropy(logits.float().reshape(-1, logits.shape[-1]), tgt.reshape(-1))
        loss.backward()
        grads = {n: p.grad.clone() for n, p in m.named_parameters()}
        outs.append((logits.detach(), loss.item(), grads))
    ref_logits, ref_loss, ref_grads = outs[0]
    worst = 0.0
    for name, (logits, loss, grads) in zip(("ConvLMHeadModel", "HyenaDNALM"), outs[1:]):
        err = ((logits - ref_logits).norm() / ref_logits.norm()).item()
        assert err < 2e-5, (name, err)
        assert abs(loss - ref_loss) < 1e-5 * abs(ref_loss) + 1e-6, (name, loss, ref_loss)
        assert set(grads) == set(ref_grads)
        for n, g in grads.items():
            e = ((g - ref_grads[n]).norm() / ref_grads[n].norm().clamp_min(1e-20)).item()
            worst = max(worst, e, err)
            assert e < 5e-4, (name, n, e)
    print(f"LM_OK params={sum(p.numel() for p in conv.parameters())} worst_rel={worst:.2e}", flush=True)


if __name__ == "__main__":
    main()
