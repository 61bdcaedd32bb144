"""A Lightning-free, Hydra-free runner for the reference's hg38 pre-training experiment (SURVEY.md 8f-1 ii / iii).

What ``python -m train experiment=hg38/hg38_hyena`` does in the reference (train.py + PyTorch-Lightning + Hydra/OmegaConf, none
of which exist in this image) reduced to what the hot path needs to be trained:

* ``compose`` -- reads the reference's OWN yaml files (``configs/config.yaml``, ``configs/experiment/hg38/hg38_hyena.yaml`` and
  the group files its ``defaults`` lists pull in: pipeline, trainer, loader, dataset, optimizer, scheduler, callbacks) with the
  composition rules Hydra applies to them (``# @package _global_``, group packages, ``override /group: name``, self last) and
  resolves OmegaConf interpolations incl. the two resolvers train.py registers (train.py:37-38: ``eval``, ``div_up``);
* ``gpu_mem_gb`` -- ``train.gpu_mem`` shells out to ``nvidia-smi`` (hg38_hyena.yaml:73); here the figure comes from the HIP runtime
  (``torch.cuda.mem_get_info``), same unit (MiB / 1000, rounded);
* ``set_affinity`` -- src/callbacks/gpu_affinity.py binds the rank to the cores of its GPU's socket via NVML / libcudart; here: the
  NUMA node of the GPU's PCI device from sysfs -> ``os.sched_setaffinity`` (no-op where sysfs has no answer);
* ``build_model`` -> ``hyena_dna_amd.lm.HyenaDNALM`` from the ``model:`` node;
* ``build_optimizer`` -- the parameter groups of ``SequenceLightningModule.configure_optimizers`` (train.py:443-468): all
  parameters without an ``_optim`` tag in the first group with the optimizer's hyperparameters, then one group per distinct
  ``_optim`` dict (the Hyena filter's ``lr`` / ``weight_decay = 0``, src/utils/train.py:142-156);
* ``TimmCosineSchedule`` -- ``scheduler: cosine_warmup_timm`` = ``timm.scheduler.CosineLRScheduler`` stepped per update
  (src/utils/optim/schedulers.py:66-87; timm is not installed, its formula is restated);
* ``train`` -- HG38Dataset batches -> bf16 autocast forward, cross entropy over the flattened logits (src/tasks/metrics.py:180-183),
  backward, gradient clipping (``trainer.gradient_clip_val``), accumulation (``trainer.accumulate_grad_batches``), AdamW, schedule;
  optionally the whole step as one hipGraph (``lm.GraphedTrainStep``); for N > 1 ranks DDP with the reference's settings
  (train.py:611-620).

Not reproduced (out of scope): Lightning's logging / checkpoint callbacks, wandb, EMA, validation
epochs, fault-tolerant samplers.
"""
import math
import contextlib
import os
import re
import time

import torch


def _token_cross_entropy():
    from .lm import token_cross_entropy          # (imported late, like the other model pieces: lm pulls the kernels' library in)
    return token_cross_entropy

__all__ = ["compose", "compose_raw", "apply_overrides", "resolve", "gpu_mem_gb", "set_affinity", "build_model", "build_optimizer", "optimizer_groups",
           "TimmCosineSchedule", "make_synthetic_genome", "train"]


# ---------------------------------------------------------------------------------------------------------------------
# config composition (Hydra's rules for the files the hg38 experiment touches)
# ---------------------------------------------------------------------------------------------------------------------
_FLOAT = re.compile(r"^[+-]?(\d+\.?\d*|\.\d+)([eE][+-]?\d+)?$")


def _retype(node):
    """PyYAML reads `6e-4` / `1e-6` as strings (YAML 1.1 wants a dot); OmegaConf reads them as floats."""
    if isinstance(node, dict):
        return {k: _retype(v) for k, v in node.items()}
    if isinstance(node, list):
        return [_retype(v) for v in node]
    if isinstance(node, str) and _FLOAT.match(node) and not node.isdigit():
        return float(node)
    return node


def _read(path):
    import yaml
    with open(path) as f:
        text = f.read()
    first = text.lstrip().splitlines()[0] if text.strip() else ""
    is_global = first.replace(" ", "").startswith("#@package_global_")
    return _retype(yaml.safe_load(text) or {}), is_global


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v if not isinstance(v, dict) else _merge({}, v)
    return dst


def _load_group(root, group, name, overrides):
    """One config file with its own defaults list merged below it; returns a dict rooted at the GLOBAL package."""
    path = os.path.join(root, group, name + ".yaml") if group else os.path.join(root, name + ".yaml")
    body, is_global = _read(path)
    defaults = body.pop("defaults", [])
    out, self_done = {}, False

    def put_self():
        if is_global or not group:
            _merge(out, body)
        else:
            node = out
            for part in group.strip("/").split("/"):
                node = node.setdefault(part, {})
            _merge(node, body)

    for d in defaults:
        if d == "_self_":
            put_self()
            self_done = True
            continue
        (key, val), = d.items()
        if key.startswith("override "):
            continue                                              # consumed by the caller (see compose)
        g = key.strip("/")
        val = overrides.get(g, val)
        for v in (val if isinstance(val, list) else [val]):
            if v is None:
                continue
            _merge(out, _load_group(root, g, str(v), overrides))
    if not self_done:
        put_self()
    return out


def compose(config_root, experiment="hg38/hg38_hyena", overrides=()):
    """``python -m train experiment=<experiment> key=value ...`` -> one resolved plain-dict config.

    config_root: the reference's ``configs/`` directory -- or a ``.json`` file written by ``compose_raw`` (the composed, still
    unresolved tree; the GPU boxes have no reference checkout).  overrides: ``["dataset.max_length=32768", ...]`` (dotted keys,
    YAML values), applied before interpolations are resolved, like Hydra's command line."""
    import copy
    import json
    if os.path.isfile(config_root):
        with open(config_root) as f:
            doc = json.load(f)
        if doc.get("experiment") != experiment:
            raise ValueError(f"{config_root} holds experiment {doc.get('experiment')!r}, not {experiment!r}")
        cfg = copy.deepcopy(doc["config"])
    else:
        cfg = compose_raw(config_root, experiment)
    return resolve(apply_overrides(cfg, overrides))


def apply_overrides(cfg, overrides):
    import yaml
    for item in overrides:
        key, _, val = item.partition("=")
        node = cfg
        parts = key.lstrip("+").split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = _retype(yaml.safe_load(val))
    return cfg


def compose_raw(config_root, experiment="hg38/hg38_hyena"):
    """The composed tree with its interpolations still in place (what Hydra holds before OmegaConf resolves)."""
    exp_body, _ = _read(os.path.join(config_root, "experiment", experiment + ".yaml"))
    group_over = {}
    for d in exp_body.get("defaults", []):
        if isinstance(d, dict):
            (key, val), = d.items()
            if key.startswith("override "):
                group_over[key[len("override "):].strip("/")] = val
    root_body, _ = _read(os.path.join(config_root, "config.yaml"))
    root_body.pop("defaults", None)
    root_body.pop("hydra", None)                                  # Hydra's own run-directory node
    cfg = _merge({}, root_body)                                   # config.yaml lists `_self_` first: everything else wins over it
    _merge(cfg, _load_group(config_root, "experiment", experiment, group_over))
    cfg.pop("experiment", None)
    return cfg


_INNER = re.compile(r"\$\{([^${}]*)\}")


def gpu_mem_gb(device=None):
    """``train.gpu_mem`` (hg38_hyena.yaml:73 asks nvidia-smi for memory.total in MiB, / 1000, rounded): the same figure from the
    HIP runtime; 0 without a ROCm device."""
    if not torch.cuda.is_available():
        return 0
    total = torch.cuda.mem_get_info(device)[1]
    return round(total / 2 ** 20 / 1000)


def _lookup(cfg, path, here):
    """absolute `a.b.c` or relative `.x` / `..x` (relative to the node that holds the key being resolved)"""
    if path.startswith("."):
        up = len(path) - len(path.lstrip("."))
        base = list(here[:len(here) - up])
        parts = base + [p for p in path.lstrip(".").split(".") if p]
    else:
        parts = path.split(".")
    node = cfg
    for p in parts:
        node = node[p]
    return node, parts


def _resolve_value(cfg, val, here, depth=0):
    if depth > 50:
        raise RecursionError(f"interpolation cycle at {'.'.join(here)}")
    while isinstance(val, str) and "${" in val:
        m = _INNER.search(val)
        if m is None:
            break
        expr = m.group(1)
        if expr.startswith("eval:"):
            code = expr[5:].strip()
            if len(code) >= 2 and code[0] == code[-1] and code[0] in "\"'":
                code = code[1:-1]
            res = gpu_mem_gb() if "nvidia-smi" in code else eval(code)            # noqa: S307 -- the reference's own resolver
        elif expr.startswith("div_up:"):
            a, b = (float(x) if "." in x or "e" in x.lower() else int(x) for x in (s.strip() for s in expr[7:].split(",")))
            res = (a + b - 1) // b
        elif expr.startswith("now:") or expr.startswith("oc.") or expr.startswith("hydra:"):
            res = ""
        else:
            target, parts = _lookup(cfg, expr.strip(), here[:-1] + [""])
            res = _resolve_value(cfg, target, parts, depth + 1)
        if m.start() == 0 and m.end() == len(val):
            val = res
        else:
            val = val[:m.start()] + str(res) + val[m.end():]
    return val


def resolve(cfg):
    """OmegaConf.resolve for plain dicts: `${a.b}`, `${.rel}`, `${eval:...}`, `${div_up:x, y}` (train.py:37-38), nested."""
    def walk(node, here):
        if isinstance(node, dict):
            for k in list(node):
                node[k] = walk(node[k], here + [k])
            return node
        if isinstance(node, list):
            return [walk(v, here + [str(i)]) for i, v in enumerate(node)]
        return _resolve_value(cfg, node, here)
    return walk(cfg, [])


def set_affinity(local_rank):
    """src/callbacks/gpu_affinity.py (NVML socket affinity) for ROCm: bind this process to the cores of the NUMA node its GPU's
    PCI device hangs off.  Returns the core set, or None when sysfs has no answer (single-socket hosts, containers)."""
    try:
        props = torch.cuda.get_device_properties(local_rank)
        bus = f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bus}/numa_node") as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            cores = set()
            for part in f.read().strip().split(","):
                lo, _, hi = part.partition("-")
                cores.update(range(int(lo), int(hi or lo) + 1))
        os.sched_setaffinity(0, cores)
        return cores
    except (OSError, AttributeError, ValueError, RuntimeError):
        return None


# ---------------------------------------------------------------------------------------------------------------------
# model, optimizer, schedule
# ---------------------------------------------------------------------------------------------------------------------
def build_model(cfg):
    """``model:`` node (``_name_: lm`` -> registry.model['lm'] = ConvLMHeadModel in the reference) -> HyenaDNALM."""
    from .lm import HyenaDNALM
    m = dict(cfg["model"])
    name = m.pop("_name_", "lm")
    if name != "lm":
        raise NotImplementedError(f"model._name_={name!r}: this runner builds the hg38 language model ('lm') only")
    layer = dict(m.pop("layer"))
    if layer.get("_name_", "hyena") != "hyena":
        raise NotImplementedError(f"model.layer._name_={layer.get('_name_')!r}: Hyena mixers only")
    return HyenaDNALM(layer=layer, **m)


def optimizer_groups(named_params, opt_cfg):
    """train.py:443-468 as data: [(hyperparameters, [parameter names])].  Group 0 = every parameter without an `_optim` tag with
    the optimizer's own settings; then one group per distinct `_optim` dict, in the reference's order
    (``sorted(list(dict.fromkeys(frozenset(hp.items()) ...)))``), its keys laid over the optimizer's."""
    named = list(named_params)
    base = {k: v for k, v in opt_cfg.items() if k != "_name_"}
    groups = [(dict(base), [n for n, p in named if not hasattr(p, "_optim")])]
    hps = [getattr(p, "_optim") for _, p in named if hasattr(p, "_optim")]
    hps = [dict(s) for s in sorted(list(dict.fromkeys(frozenset(hp.items()) for hp in hps)))]
    for hp in hps:
        groups.append(({**base, **hp}, [n for n, p in named if getattr(p, "_optim", None) == hp]))
    return groups


def build_optimizer(model, opt_cfg, capturable=False):
    """AdamW (registry.optimizer['adamw'] = torch.optim.AdamW) over the groups of `optimizer_groups`."""
    name = opt_cfg.get("_name_", "adamw")
    if name != "adamw":
        raise NotImplementedError(f"optimizer._name_={name!r}: the hg38 experiments use adamw")
    params = dict(model.named_parameters())
    groups = []
    for hp, names in optimizer_groups(params.items(), opt_cfg):
        hp = dict(hp)
        if "betas" in hp:
            hp["betas"] = tuple(hp["betas"])
        groups.append({"params": [params[n] for n in names], **hp})
    if capturable:
        # inside a hipGraph a Python-float learning rate is frozen at its value at capture time: keep it in a device tensor that the
        # schedule updates in place between replays (lm.GraphedTrainStep)
        dev = groups[0]["params"][0].device
        for g in groups:
            g["lr"] = torch.tensor(float(g["lr"]), dtype=torch.float32, device=dev)
    first = {k: v for k, v in groups[0].items() if k != "params"}
    extra = {"capturable": True} if capturable else {}
    opt = torch.optim.AdamW(groups[0]["params"], **first, **extra)
    for g in groups[1:]:
        opt.add_param_group(g)
    return opt


class TimmCosineSchedule:
    """``timm.scheduler.CosineLRScheduler(optimizer, t_initial, lr_min, warmup_t, warmup_lr_init, t_in_epochs=False)`` with
    timm's defaults (one cycle, no warm-up prefix, k_decay 1) stepped once per optimizer update, as
    ``TimmCosineLRScheduler`` does (src/utils/optim/schedulers.py:66-87: ``step()`` -> ``step_update(num_updates)``;
    the constructor already applies update 0):

        t <  warmup_t : lr = warmup_lr_init + t (base_lr - warmup_lr_init) / warmup_t
        t >= warmup_t : lr = lr_min + (base_lr - lr_min) (1 + cos(pi t / t_initial)) / 2      for t < t_initial, lr_min after

    per parameter group, `base_lr` = the group's learning rate at construction."""

    def __init__(self, optimizer, t_initial, lr_min=0.0, warmup_t=0, warmup_lr_init=0.0, **unused):
        self.optimizer = optimizer
        self.t_initial, self.lr_min, self.warmup_t, self.warmup_lr_init = int(t_initial), float(lr_min), float(warmup_t), float(warmup_lr_init)
        self.base = [float(g["lr"]) if not torch.is_tensor(g["lr"]) else float(g["lr"].item()) for g in optimizer.param_groups]
        self.t = -1
        self.step(0)

    def lrs(self, t):
        if t < self.warmup_t:
            return [self.warmup_lr_init + t * (b - self.warmup_lr_init) / self.warmup_t for b in self.base]
        if t < self.t_initial:
            return [self.lr_min + 0.5 * (b - self.lr_min) * (1 + math.cos(math.pi * t / self.t_initial)) for b in self.base]
        return [self.lr_min for _ in self.base]

    def step(self, t=None):
        self.t = self.t + 1 if t is None else int(t)
        for g, lr in zip(self.optimizer.param_groups, self.lrs(self.t)):
            if torch.is_tensor(g["lr"]):
                g["lr"].fill_(lr)                                   # capturable optimizers keep lr on the device
            else:
                g["lr"] = lr


def build_scheduler(optimizer, sch_cfg):
    name = sch_cfg.get("_name_", "cosine_warmup_timm")
    if name != "cosine_warmup_timm":
        raise NotImplementedError(f"scheduler._name_={name!r}: the hg38 experiment overrides it to cosine_warmup_timm")
    kw = {k: v for k, v in sch_cfg.items() if k not in ("_name_", "t_in_epochs")}
    return TimmCosineSchedule(optimizer, **kw)


# ---------------------------------------------------------------------------------------------------------------------
# data
# ---------------------------------------------------------------------------------------------------------------------
def make_synthetic_genome(directory, n_chr=2, chr_len=400_000, n_intervals=64, interval_len=32768, seed=0):
    """A FASTA + a BED file in the layout the hg38 loader expects (hg38_dataset.py:123-160: columns chr, start, end, split) with a
    learnable structure (a noisy periodic motif), for smoke runs where hg38.ml.fa does not exist.  Returns (fasta, bed)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    os.makedirs(directory, exist_ok=True)
    fasta, bed = os.path.join(directory, "synthetic.fa"), os.path.join(directory, "synthetic.bed")
    motif = rng.integers(0, 4, 24)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    with open(fasta, "wb") as f:
        for c in range(n_chr):
            seq = np.tile(motif, chr_len // motif.size + 1)[:chr_len].copy()
            flip = rng.random(chr_len) < 0.05
            seq[flip] = rng.integers(0, 4, int(flip.sum()))
            body = letters[seq]
            f.write(f">chr{c + 1}\n".encode())
            for i in range(0, chr_len, 60):
                f.write(body[i:i + 60].tobytes() + b"\n")
    with open(bed, "w") as f:
        for i in range(n_intervals):
            c = int(rng.integers(0, n_chr)) + 1
            start = int(rng.integers(0, max(1, chr_len - interval_len)))
            split = "train" if i % 8 < 6 else ("valid" if i % 8 == 6 else "test")
            f.write(f"chr{c}\t{start}\t{start + interval_len}\t{split}\n")
    return fasta, bed


def build_dataset(cfg, split="train"):
    from .dataset import HG38Dataset
    d = cfg["dataset"]
    max_len = {"train": d["max_length"], "valid": d.get("max_length_val") or d["max_length"],
               "test": d.get("max_length_test") or d["max_length"]}[split]
    # src/dataloaders/genomics.py:78-82: unset paths fall back to <data dir>/hg38/{human-sequences.bed, hg38.ml.fa}
    data_dir = os.environ.get("DATA_PATH") or os.path.join(os.getcwd(), "data")
    bed = d.get("bed_file") or os.path.join(data_dir, "hg38", "human-sequences.bed")
    fasta = d.get("fasta_file") or os.path.join(data_dir, "hg38", "hg38.ml.fa")
    for what, path, key in (("interval list", bed, "dataset.bed_file"), ("genome", fasta, "dataset.fasta_file")):
        if not os.path.exists(path):
            raise FileNotFoundError(f"hg38 {what} not found at {path!r}: pass {key}=<path> (the reference's default location is "
                                    f"data/hg38/, src/dataloaders/genomics.py:78-82), or use --synthetic-genome")
    # src/dataloaders/genomics.py:127-141
    return HG38Dataset(split=split, bed_file=bed, fasta_file=fasta, max_length=max_len, tokenizer=None,
                       tokenizer_name=d.get("tokenizer_name") or "char", add_eos=d.get("add_eos", True), return_seq_indices=False,
                       shift_augs=None, rc_aug=d.get("rc_aug", False), return_augs=False,
                       replace_N_token=d.get("replace_N_token", False), pad_interval=d.get("pad_interval", False))


def precision_dtype(precision):
    """trainer.precision -> the autocast type (None: no autocast).  Lightning 1.8.6 spellings: 16 / "16" / "16-mixed" -> float16 (with a loss
    scaler, see train), "bf16" / "bf16-mixed" -> bfloat16, 32 / "32" / "32-true" -> None.  Lightning's other spellings are refused BY NAME: "64" /
    "64-true" (this package's kernels compute in fp32 or 16 bits) and the "16-true" / "bf16-true" of Lightning 2 (16-bit PARAMETERS: the reference's
    pinned Lightning 1.8.6 has no such mode and the fused optimizer path keeps fp32 master weights)."""
    p = str(precision).strip().lower()
    if p in ("16", "16-mixed"):
        return torch.float16
    if p in ("bf16", "bf16-mixed"):
        return torch.bfloat16
    if p in ("32", "32-true"):
        return None
    if p in ("64", "64-true"):
        raise ValueError(f"trainer.precision={precision!r}: double precision is not served (kernels compute in fp32 or 16 bits); use 32")
    if p in ("16-true", "bf16-true"):
        raise ValueError(f"trainer.precision={precision!r}: 16-bit parameters (Lightning 2's '-true' modes) are not served; the mixed modes "
                         f"16 / 16-mixed / bf16 / bf16-mixed keep fp32 master weights as the reference's Lightning 1.8.6 does")
    raise ValueError(f"trainer.precision={precision!r}: expected one of 16, 16-mixed, bf16, bf16-mixed, 32, 32-true")


# ---------------------------------------------------------------------------------------------------------------------
# the loop
# ---------------------------------------------------------------------------------------------------------------------
def train(cfg, max_steps, device, graphed=False, log_every=10, log=print):
    """Runs `max_steps` optimizer updates of the experiment `cfg` describes; returns the list of per-update losses."""
    import random as _random
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    seed = int(cfg.get("train", {}).get("seed", 0))
    torch.manual_seed(seed)                                       # pl.seed_everything(config.train.seed) (train.py:673-674)
    _random.seed(seed + rank)
    model = build_model(cfg).to(device)
    net = model
    if world > 1:
        from torch.nn.parallel import DistributedDataParallel
        net = DistributedDataParallel(model, device_ids=[device.index] if device.type == "cuda" else None,
                                      find_unused_parameters=False, gradient_as_bucket_view=True)       # train.py:611-620
    opt = build_optimizer(model, cfg["optimizer"], capturable=graphed)
    sched = build_scheduler(opt, cfg["scheduler"]) if "scheduler" in cfg else None
    ds = build_dataset(cfg, "train")
    sampler = torch.utils.data.distributed.DistributedSampler(ds, shuffle=cfg["dataset"].get("shuffle", True), seed=seed) if world > 1 else None
    loader = torch.utils.data.DataLoader(ds, batch_size=int(cfg["dataset"]["batch_size"]), sampler=sampler,
                                         shuffle=(sampler is None and bool(cfg["dataset"].get("shuffle", True))),
                                         drop_last=bool(cfg.get("loader", {}).get("drop_last", True)),
                                         num_workers=int(cfg["dataset"].get("num_workers", 0) or 0))
    tr = cfg.get("trainer", {})
    accum = max(1, int(tr.get("accumulate_grad_batches", 1) or 1))
    clip = float(tr.get("gradient_clip_val", 0.0) or 0.0)
    # trainer.precision (hg38_hyena.yaml:41 `precision: 16`; PyTorch Lightning 1.8.6's native AMP plugin): 16 = float16 autocast WITH a
    # dynamic loss scaler (torch.amp.GradScaler, PyTorch's defaults: 65536, x2 every 2000 clean steps, / 2 and the update skipped on an inf / nan
    # gradient; gradients are unscaled before clipping, as Lightning does); bf16 = bfloat16 autocast, no scaler (the yaml's own comment:
    # "bf16 only a100" -- what the MI355X benchmarks of this repository run); 32 = no autocast.  Round 5: until then 16 was mapped to bf16.
    amp_dtype = precision_dtype(tr.get("precision", 32))
    amp = amp_dtype is not None
    dev_type = device.type
    scaler = torch.amp.GradScaler(dev_type, enabled=True) if (amp_dtype == torch.float16 and dev_type == "cuda") else None
    log(f"[runner] params {sum(p.numel() for p in model.parameters())}, groups "
        f"{[(len(g['params']), g['lr'] if not torch.is_tensor(g['lr']) else float(g['lr']), g['weight_decay']) for g in opt.param_groups]}, "
        f"accumulate {accum}, clip {clip}, gpu_mem {cfg.get('train', {}).get('gpu_mem')}")

    def batches():
        epoch = 0
        while True:
            if sampler is not None:
                sampler.set_epoch(epoch)
            for x, y in loader:
                yield x.to(device, non_blocking=True), y.to(device, non_blocking=True)
            epoch += 1

    it = batches()
    losses, t0 = [], time.perf_counter()
    if graphed:
        if accum != 1 or world != 1:
            raise NotImplementedError("graphed=True captures one micro-batch per update on one GPU")
        if scaler is not None:
            raise NotImplementedError("graphed=True with trainer.precision=16: the loss scaler's skipped updates are decided on the host; "
                                      "capture bf16 (trainer.precision=bf16) or fp32 steps")
        from .lm import GraphedTrainStep
        x, y = next(it)
        # the capture's warm-up updates are undone (parameters, moments, step counters: lm.GraphedTrainStep), and the batch it warmed up on
        # is the first counted step's batch: a graphed and an eager run of the same config + seed follow the same trajectory
        step = GraphedTrainStep(model, opt, x, y, autocast_dtype=amp_dtype, warmup=2, clip_grad_norm=clip)
        for i in range(max_steps):
            if i > 0:
                x, y = next(it)
            loss = step(x, y)
            if sched is not None:
                sched.step()
            losses.append(float(loss))
            if (i + 1) % log_every == 0:
                log(f"[runner] step {i + 1} loss {losses[-1]:.4f} ({(time.perf_counter() - t0) / (i + 1) * 1e3:.1f} ms/step, graphed)")
        return losses
    for i in range(max_steps):
        opt.zero_grad(set_to_none=True)
        total = 0.0
        for a in range(accum):
            x, y = next(it)
            # gradients cross the ranks once per update, on the last micro-batch (what Lightning does with accumulate_grad_batches;
            # hg38_hyena resolves it to >= 64, so all-reducing every micro-batch would multiply the collective traffic by that)
            hold = net.no_sync() if (world > 1 and a + 1 < accum) else contextlib.nullcontext()
            with hold:
                with torch.autocast(dev_type, dtype=amp_dtype or torch.bfloat16, enabled=amp and dev_type == "cuda"):
                    logits = net(x)[0].logits
                    loss = _token_cross_entropy()(logits, y)
                micro = loss / accum
                (scaler.scale(micro) if scaler is not None else micro).backward()
            total += float(loss.detach()) / accum
        if scaler is not None:
            if clip > 0:
                scaler.unscale_(opt)                              # clip the TRUE gradients (Lightning: precision_plugin.pre_optimizer_step -> unscale, then clip)
                torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
            scaler.step(opt)                                      # skipped when a gradient overflowed in fp16
            scaler.update()
        else:
            if clip > 0:
                torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
            opt.step()
        if sched is not None:
            sched.step()
        losses.append(total)
        if (i + 1) % log_every == 0:
            log(f"[runner] step {i + 1} loss {total:.4f} lr {opt.param_groups[0]['lr']:.3e} "
                f"({(time.perf_counter() - t0) / (i + 1) * 1e3:.1f} ms/step)" + (f" loss scale {scaler.get_scale():.0f}" if scaler is not None else ""))
    return losses
