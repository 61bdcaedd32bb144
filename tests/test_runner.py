"""hyena_dna_amd.runner (SURVEY.md 8f-1 ii/iii): the Hydra-free composition of the reference's hg38 experiment config, the
optimizer parameter groups of train.py:443-468, the timm cosine schedule of `cosine_warmup_timm`, and a few training steps on a
synthetic genome (kernels under tests/hipemu)."""
import json
import math
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("HYENA_REFERENCE", "/root/reference")
COMPOSED = os.path.join(ROOT, "tests", "golden", "hg38_hyena_composed.json")
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "configs")), reason="reference checkout not present")


def test_experiment_config_resolves_like_hydra():
    from hyena_dna_amd import runner
    cfg = runner.compose(COMPOSED)
    # values a Hydra run of `experiment=hg38/hg38_hyena` prints (hg38_hyena.yaml + pipeline/hg38.yaml + group files)
    assert cfg["model"]["d_inner"] == 4 * cfg["model"]["d_model"] == 128 and cfg["model"]["layer"]["l_max"] == 1026
    assert cfg["model"]["layer"]["lr"] == cfg["optimizer"]["lr"] == 6e-4 and cfg["optimizer"]["weight_decay"] == 0.1
    assert cfg["optimizer"]["betas"] == [0.9, 0.999] and cfg["optimizer"]["_name_"] == "adamw"
    assert cfg["dataset"]["batch_size"] == 256 and cfg["dataset"]["batch_size_eval"] == 512 and cfg["dataset"]["max_length_val"] == 1024
    assert cfg["dataset"]["__train_len"] == (10 ** 9 + 1023) // 1024 == 976563
    steps_per_epoch = (976563 + 255) // 256
    assert cfg["trainer"]["accumulate_grad_batches"] == 1 and cfg["trainer"]["gradient_clip_val"] == 1.0
    assert cfg["scheduler"] == {"_name_": "cosine_warmup_timm", "t_in_epochs": False, "t_initial": steps_per_epoch * 100,
                                "lr_min": pytest.approx(6e-5), "warmup_lr_init": 1e-6, "warmup_t": pytest.approx(steps_per_epoch)}
    assert cfg["train"]["interval"] == "step" and cfg["train"]["seed"] == 2222 and cfg["train"]["monitor"] == "test/loss"
    assert cfg["train"]["gpu_mem"] == 0                    # no ROCm device here: the nvidia-smi call is replaced, not executed
    # overrides act before resolution, like Hydra's command line
    big = runner.compose(COMPOSED, overrides=["dataset.max_length=1048576", "dataset.batch_size=1", "trainer.devices=8",
                                               "model.d_model=256", "model.n_layer=8"])
    assert big["model"]["layer"]["l_max"] == 1048578 and big["model"]["d_inner"] == 1024
    assert big["trainer"]["accumulate_grad_batches"] == 256 // 8


def _flatten(tree, prefix=""):
    out = {}
    for k, v in tree.items():
        if isinstance(v, dict):
            out.update(_flatten(v, prefix + k + "."))
        else:
            out[prefix + k] = v
    return out


def test_resolved_config_equals_the_hand_derived_tree_key_by_key():
    """the composition used to be pinned only to a file the runner itself wrote.  tests/golden/hg38_hyena_resolved_by_hand.py
    is the resolved tree derived by hand from the reference's yaml files with Hydra's merge rules (its docstring lists them): every key and
    every value -- including the types (3815.0 is a float, 381500 an int) -- must agree, and neither side may have a key the other lacks."""
    import importlib.util
    from hyena_dna_amd import runner
    spec = importlib.util.spec_from_file_location("by_hand", os.path.join(ROOT, "tests", "golden", "hg38_hyena_resolved_by_hand.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = _flatten(mod.RESOLVED)
    got = _flatten(runner.compose(COMPOSED))
    gpu_mem = got.pop("train.gpu_mem")                    # run-time (nvidia-smi in the reference, the HIP runtime here)
    assert isinstance(gpu_mem, int)
    assert sorted(got) == sorted(want), (sorted(set(got) - set(want)), sorted(set(want) - set(got)))
    for key in want:
        assert got[key] == want[key] and type(got[key]) is type(want[key]), (key, got[key], want[key])


@needs_ref
def test_composed_json_is_what_the_reference_configs_compose_to():
    from hyena_dna_amd import runner
    with open(COMPOSED) as f:
        doc = json.load(f)
    assert runner.compose_raw(os.path.join(REF, "configs"), "hg38/hg38_hyena") == doc["config"]


def test_timm_cosine_schedule_formula():
    from hyena_dna_amd.runner import TimmCosineSchedule
    p = [torch.nn.Parameter(torch.zeros(1)), torch.nn.Parameter(torch.zeros(1))]
    opt = torch.optim.AdamW([{"params": [p[0]], "lr": 6e-4}, {"params": [p[1]], "lr": 1e-3}])
    s = TimmCosineSchedule(opt, t_initial=100, lr_min=6e-5, warmup_t=10, warmup_lr_init=1e-6)
    assert opt.param_groups[0]["lr"] == pytest.approx(1e-6)           # update 0 applied by the constructor
    for t in range(1, 130):
        s.step()
        for g, base in zip(opt.param_groups, (6e-4, 1e-3)):
            if t < 10:
                want = 1e-6 + t * (base - 1e-6) / 10
            elif t < 100:
                want = 6e-5 + 0.5 * (base - 6e-5) * (1 + math.cos(math.pi * t / 100))
            else:
                want = 6e-5
            assert g["lr"] == pytest.approx(want, rel=1e-12)


_GROUPS_WORKER = r'''
import importlib, json, os, sys, types
import torch
ROOT, REF = sys.argv[1], sys.argv[2]
sys.path[:0] = [ROOT, REF]
def _stub(name, **attrs):
    m = types.ModuleType(name); m.__dict__.update(attrs); sys.modules[name] = m; return m
def _get(path):
    mod, _, attr = path.rpartition("."); return getattr(importlib.import_module(mod), attr)
_stub("hydra", utils=_stub("hydra.utils", get_method=_get, get_class=_get))
_stub("omegaconf", ListConfig=list, DictConfig=type("DictConfig", (dict,), {}), OmegaConf=object)
_stub("pytorch_lightning", utilities=_stub("pytorch_lightning.utilities", rank_zero_only=lambda f: f))
_stub("opt_einsum", contract=torch.einsum)
import transformers.tokenization_utils
class _SD(torch.nn.Module):
    def __init__(self, p, mode): super().__init__()
    def forward(self, x): return x
_stub("torchvision", ops=_stub("torchvision.ops", StochasticDepth=_SD))
import src.models.sequence.simple_lm as ref_simple
from hyena_dna_amd import runner
cfg = runner.compose(os.path.join(REF, "configs"), overrides=["model.d_model=64", "optimizer.lr=3e-4"])
m = dict(cfg["model"]); m.pop("_name_"); layer = dict(m.pop("layer"))
for k in ("fused_mlp", "fused_dropout_add_ln", "checkpoint_mixer", "checkpoint_mlp"): m.pop(k)
ref_model = ref_simple.SimpleLMHeadModel(layer=layer, **m)
# --- train.py:443-468, the reference's lines on the reference's model (hparams.optimizer = cfg["optimizer"]) ---
hp_opt = dict(cfg["optimizer"]); hp_opt.pop("_name_")
all_params = list(ref_model.named_parameters())
groups = [({k: v for k, v in hp_opt.items()}, [n for n, p in all_params if not hasattr(p, "_optim")])]
hps = [getattr(p, "_optim") for _, p in all_params if hasattr(p, "_optim")]
hps = [dict(s) for s in sorted(list(dict.fromkeys(frozenset(hp.items()) for hp in hps)))]
for hp in hps:
    groups.append(({**hp_opt, **hp}, [n for n, p in all_params if getattr(p, "_optim", None) == hp]))
# --- this package's model and runner ---
mine = runner.build_model(cfg)
opt = runner.build_optimizer(mine, cfg["optimizer"])
names = {id(p): n for n, p in mine.named_parameters()}
got = [({k: (list(v) if isinstance(v, tuple) else v) for k, v in g.items() if k in want_hp}, [names[id(p)] for p in g["params"]])
       for g, (want_hp, _) in zip(opt.param_groups, groups)]
assert len(opt.param_groups) == len(groups) == 2, (len(opt.param_groups), len(groups))
for (hp_g, names_g), (hp_r, names_r) in zip(got, groups):
    assert hp_g == hp_r, (hp_g, hp_r)
    assert names_g == names_r, (set(names_g) ^ set(names_r))
assert groups[1][0]["lr"] == 3e-4 and groups[1][0]["weight_decay"] == 0.0 and groups[0][0]["weight_decay"] == 0.1
print("GROUPS_OK", [len(n) for _, n in groups])
'''


@needs_ref
def test_optimizer_groups_equal_the_reference_configure_optimizers(tmp_path):
    """The reference's group-building lines (train.py:443-468) run on the reference's own model (SimpleLMHeadModel with the
    reference HyenaOperator and its `_optim` tags) give the same groups -- hyperparameters, membership AND order of the
    parameters -- as runner.build_optimizer on HyenaDNALM."""
    env = dict(os.environ, OMP_NUM_THREADS="2")
    env.pop("PYTHONPATH", None)
    p = subprocess.run([sys.executable, "-c", _GROUPS_WORKER, ROOT, REF], cwd=str(tmp_path), env=env, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-2500:])
    assert "GROUPS_OK" in p.stdout


def test_runner_trains_on_a_synthetic_genome(tmp_path):
    """scripts/train_hg38.py end to end on the emulated kernels: config -> model -> groups -> HG38Dataset batches -> steps"""
    env = dict(os.environ, OMP_NUM_THREADS="4")
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "train_hg38.py"), "--configs", COMPOSED, "--emu", "--steps", "12",
           "--synthetic-genome", str(tmp_path / "genome"), "dataset.max_length=256", "dataset.batch_size=4", "model.d_model=64",
           "model.fused_dropout_add_ln=true", "model.embed_dropout=0.0", "scheduler.warmup_t=2", "scheduler.t_initial=40",
           "optimizer.lr=3e-3", "trainer.accumulate_grad_batches=2", "train.global_batch_size=8"]
    p = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-2500:])
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert r["steps"] == 12 and r["fell"] and all(l == l for l in r["losses"]), r


def test_precision_spellings():
    """trainer.precision: Lightning 1.8.6's mixed modes are served, the others refused by name"""
    import torch
    from hyena_dna_amd import runner
    assert runner.precision_dtype(16) == torch.float16 and runner.precision_dtype("16-mixed") == torch.float16
    assert runner.precision_dtype("bf16") == torch.bfloat16 and runner.precision_dtype("BF16-mixed") == torch.bfloat16
    assert runner.precision_dtype(32) is None and runner.precision_dtype("32-true") is None
    for bad, word in ((64, "double"), ("16-true", "16-bit parameters"), ("bf16-true", "16-bit parameters"), ("fp8", "expected one of")):
        with pytest.raises(ValueError, match=word):
            runner.precision_dtype(bad)
