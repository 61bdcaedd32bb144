"""SURVEY.md 8f-1 (ii)/(iii) on the MI355X: hyena_dna_amd.runner.train -- the reference's hg38_hyena experiment (its own yaml files,
composed without Hydra / Lightning) -- for a few optimizer updates on a synthetic genome through the HIP kernels: the loss falls, the
optimizer groups are those of train.py:443-468, and the hipGraph-replayed loop follows the eager one."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMPOSED = os.path.join(ROOT, "tests", "golden", "hg38_hyena_composed.json")
OVERRIDES = ["dataset.max_length=8192", "dataset.batch_size=2", "model.d_model=128", "model.n_layer=2", "model.fused_dropout_add_ln=true",
             "model.embed_dropout=0.0", "scheduler.warmup_t=2", "scheduler.t_initial=60", "optimizer.lr=3e-3",
             "train.global_batch_size=2", "trainer.precision=bf16", "dataset.num_workers=0"]


@pytest.fixture(scope="module")
def genome(tmp_path_factory):
    from hyena_dna_amd import runner
    d = tmp_path_factory.mktemp("genome")
    fasta, bed = runner.make_synthetic_genome(str(d), chr_len=4 * 8192 + 1000, interval_len=8192)
    return [f"dataset.fasta_file={fasta}", f"dataset.bed_file={bed}"]


def test_runner_trains_hg38_hyena_on_the_gpu(gpu_lib, genome):
    from hyena_dna_amd import _lib, runner
    dev = torch.device("cuda", 0)
    cfg = runner.compose(COMPOSED, overrides=OVERRIDES + genome)
    assert cfg["model"]["layer"]["l_max"] == 8194 and cfg["trainer"]["accumulate_grad_batches"] == 1
    # the model on the device takes the HIP paths: fused filter (16-bit autocast kernels), matrix-core in_proj, long conv
    calls = []
    real_f, real_c = _lib.filter_fwd, _lib.fftconv_fwd
    _lib.filter_fwd = lambda *a, **k: (calls.append("filter"), real_f(*a, **k))[1]
    _lib.fftconv_fwd = lambda *a, **k: (calls.append("fftconv"), real_c(*a, **k))[1]
    logs = []
    try:
        losses = runner.train(cfg, 12, dev, graphed=False, log_every=4, log=logs.append)
    finally:
        _lib.filter_fwd, _lib.fftconv_fwd = real_f, real_c
    assert "filter" in calls and "fftconv" in calls
    assert len(losses) == 12 and all(l == l for l in losses)
    assert sum(losses[-3:]) / 3 < 0.97 * sum(losses[:3]) / 3, losses              # a 4-letter genome with structure: the loss falls
    assert losses[0] < 3.0                                                       # ~ln(16) at random init, never exploded
    # the optimizer groups the runner reported: train.py:443-468 -- the untagged parameters with the optimizer's hyperparameters
    # (weight decay 0.1), then ONE group for the `_optim`-tagged filter parameters (lr = optimizer.lr, wd 0.0: hg38_hyena.yaml:24-27)
    head = [l for l in logs if "groups" in l][0]
    assert ", 0.1)" in head and ", 0.0)" in head, head
    model = runner.build_model(cfg)
    opt = runner.build_optimizer(model, cfg["optimizer"])
    tagged = [n for n, p in model.named_parameters() if hasattr(p, "_optim")]
    assert len(opt.param_groups) == 2 and len(opt.param_groups[1]["params"]) == len(tagged) > 0
    assert opt.param_groups[0]["weight_decay"] == 0.1 and opt.param_groups[1]["weight_decay"] == 0.0
    assert all(("filter_fn" in n) for n in tagged), tagged


def test_graphed_runner_follows_the_eager_runner(gpu_lib, genome):
    """Same config, same seed: the loop that replays one captured hipGraph per update gives the eager loop's losses (the warm-up updates
    of the capture are undone) -- dropout is off in this config, so no RNG stream is involved."""
    from hyena_dna_amd import runner
    dev = torch.device("cuda", 0)
    cfg = runner.compose(COMPOSED, overrides=OVERRIDES + genome + ["trainer.gradient_clip_val=0.0"])
    eager = runner.train(cfg, 6, dev, graphed=False, log=lambda *_: None)
    graphed = runner.train(cfg, 6, dev, graphed=True, log=lambda *_: None)
    assert all(abs(a - b) <= 5e-3 * abs(a) for a, b in zip(eager, graphed)), (eager, graphed)


def test_runner_precision_16_is_fp16_with_a_loss_scaler(gpu_lib, genome):
    """hg38_hyena.yaml:41 `precision: 16` under PyTorch Lightning 1.8.6 = float16 autocast + a dynamic loss scaler (the
    runner used to train bf16 there).  12 updates through the fp16 kernels (filter16, matrix-core projections / MLP, long conv with fp16
    rows): finite, falling loss; the scaler is live (its scale is logged and stays a power of two); the trajectory stays near the bf16 one."""
    from hyena_dna_amd import runner
    dev = torch.device("cuda", 0)
    assert runner.precision_dtype(16) == torch.float16 and runner.precision_dtype("bf16") == torch.bfloat16 and runner.precision_dtype(32) is None
    base = [o for o in OVERRIDES if not o.startswith("trainer.precision")] + genome
    logs = []
    fp16 = runner.train(runner.compose(COMPOSED, overrides=base + ["trainer.precision=16"]), 12, dev, log_every=4, log=logs.append)
    bf16 = runner.train(runner.compose(COMPOSED, overrides=base + ["trainer.precision=bf16"]), 12, dev, log=lambda *_: None)
    assert len(fp16) == 12 and all(l == l and l < 3.0 for l in fp16)
    assert sum(fp16[-3:]) / 3 < 0.97 * sum(fp16[:3]) / 3, fp16
    scales = [float(l.split("loss scale")[1]) for l in logs if "loss scale" in l]
    assert scales and all(s_ >= 1.0 and (s_ == 2.0 ** round(__import__("math").log2(s_))) for s_ in scales), logs
    assert abs(fp16[0] - bf16[0]) < 2e-2 and abs(sum(fp16[-3:]) - sum(bf16[-3:])) / 3 < 0.15, (fp16, bf16)
    with pytest.raises(NotImplementedError):
        runner.train(runner.compose(COMPOSED, overrides=base + ["trainer.precision=16"]), 2, dev, graphed=True, log=lambda *_: None)
