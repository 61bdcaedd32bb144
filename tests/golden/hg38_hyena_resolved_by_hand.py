"""The fully resolved `experiment=hg38/hg38_hyena` config tree, DERIVED BY HAND from the reference's yaml files -- not produced by
hyena_dna_amd.runner (tests/test_runner.py compares the runner's composition + resolution against this, key by key; the file the runner
itself wrote, hg38_hyena_composed.json, is only its cache of the unresolved tree for GPU boxes without the reference checkout).

Hydra's rules applied (hydra 1.x defaults list; OmegaConf resolvers `eval` = Python eval, `div_up` = (x + y - 1) // y, train.py:37-38):
  * configs/config.yaml: defaults [_self_, experiment]  ->  config.yaml's own body merges FIRST, the experiment on top of it;
  * configs/experiment/hg38/hg38_hyena.yaml (`# @package _global_`): defaults [/pipeline: hg38, override /scheduler: cosine_warmup_timm],
    no `_self_`  ->  its own body merges LAST; the `override` replaces the pipeline's scheduler CHOICE (cosine_warmup's keys never appear);
  * configs/pipeline/hg38.yaml (`_global_`): /trainer default, /loader default, /dataset hg38, /optimizer adamw, /scheduler <choice>,
    /callbacks [base, checkpoint], then its own body (train.monitor / mode, task, encoder, decoder);
  * scheduler/cosine_warmup_timm.yaml is `_global_` and carries train.interval = step next to the scheduler keys;
  * later merges win key by key, dictionaries merge recursively; interpolations resolve after the merge;
  * OmegaConf's YAML loader reads 1e-6 / 6e-4 as floats (PyYAML alone would read strings).
The `hydra:` block of config.yaml is Hydra's own and is not part of the job config.  `train.gpu_mem` is a shell-out to nvidia-smi
(hg38_hyena.yaml:72): run-time, compared separately.
"""

RESOLVED = {
    # ---- configs/config.yaml:16-50 (train), then cosine_warmup_timm.yaml:2-3 (interval), pipeline/hg38.yaml:10-12 (monitor, mode),
    #      hg38_hyena.yaml:71-74 (gpu_mem, seed, global_batch_size)
    "train": {
        "seed": 2222,                                   # config.yaml:17 says 0; hg38_hyena.yaml:73 wins
        "interval": "step",                             # ??? in config.yaml:20; scheduler/cosine_warmup_timm.yaml:3
        "monitor": "test/loss",                         # pipeline/hg38.yaml:11
        "mode": "min",                                  # pipeline/hg38.yaml:12
        "ema": 0.0, "test": False, "debug": False, "ignore_warnings": False,
        "state": {"mode": None, "n_context": 0, "n_context_eval": 0},          # ${.n_context}
        "ckpt": None, "disable_dataset": False, "validate_at_start": False,
        "pretrained_model_path": None, "pretrained_model_strict_load": True,
        "pretrained_model_state_hook": {"_name_": None}, "post_init_hook": {"_name_": None},
        "layer_decay": {"_name_": None, "decay": 0.7},
        "global_batch_size": 256,                       # hg38_hyena.yaml:74
    },
    "tolerance": {"logdir": "./resume", "id": None},                           # config.yaml:52-54
    "wandb": {"project": "dna", "group": "", "job_type": "training", "mode": "online", "name": None, "save_dir": ".",
              "id": None},                                                     # config.yaml:59-66; id = ${.name}
    # ---- trainer/default.yaml, then hg38_hyena.yaml:32-40
    "trainer": {
        "_target_": "pytorch_lightning.Trainer",
        "devices": 1, "accelerator": "gpu", "num_nodes": 1,
        "accumulate_grad_batches": 1,                   # div_up(256, eval(1 * 256 * 1)) = (256 + 255) // 256
        "max_epochs": 100,                              # default 200; hg38_hyena.yaml:37
        "gradient_clip_val": 1.0,                       # default 0.0; hg38_hyena.yaml:39
        "log_every_n_steps": 10, "limit_train_batches": 1.0, "limit_val_batches": 1.0,
        "precision": 16,
    },
    "loader": {"batch_size": 50, "num_workers": 4, "pin_memory": True, "drop_last": True},      # loader/default.yaml
    # ---- dataset/hg38.yaml, then hg38_hyena.yaml:42-57
    "dataset": {
        "_name_": "hg38", "bed_file": None, "fasta_file": None, "dataset_name": "hg38",
        "tokenizer_name": "char",                       # null in the group file; hg38_hyena.yaml:49
        "cache_dir": None,
        "max_length": 1024, "add_eos": True,
        "batch_size": 256,                              # 8 in the group file; hg38_hyena.yaml:44
        "batch_size_eval": 512,                         # eval(256 * 2)
        "num_workers": 12,                              # 4 in the group file; hg38_hyena.yaml:53
        "shuffle": True, "pin_memory": True,
        "__train_len": 976563,                          # div_up(1_000_000_000, 1024) = (10**9 + 1023) // 1024
        "__l_max": 1024,
        "max_length_val": 1024, "max_length_test": 1024,
        "pad_max_length": None, "rc_aug": False, "use_fixed_len_val": False, "replace_N_token": False, "pad_interval": False,
    },
    "optimizer": {"_name_": "adamw", "lr": 6e-4, "weight_decay": 0.1, "betas": [0.9, 0.999]},   # adamw.yaml + hg38_hyena.yaml:66-68
    # ---- scheduler/cosine_warmup_timm.yaml:4-11, then hg38_hyena.yaml:59-64.  steps per epoch = div_up(976563, 256) = 3815
    "scheduler": {
        "_name_": "cosine_warmup_timm", "t_in_epochs": False,
        "t_initial": 3815 * 100,
        "warmup_lr_init": 1e-6,
        "warmup_t": 3815 * 100 * 0.01,                  # eval("3815 * 100 * 0.01"): a float
        "lr_min": 0.1 * 0.0006,                         # eval("0.1 * 0.0006")
    },
    "callbacks": {                                      # callbacks/base.yaml + callbacks/checkpoint.yaml
        "learning_rate_monitor": {"logging_interval": "step"},
        "timer": {"step": True, "inter_step": False, "epoch": True, "val": True},
        "params": {"total": True, "trainable": True, "fixed": True},
        "model_checkpoint": {"monitor": "test/loss", "mode": "min", "save_top_k": 1, "save_last": True, "dirpath": "checkpoints/",
                             "filename": "test/loss", "auto_insert_metric_name": False, "verbose": True},
    },
    "task": {"_name_": "lm", "loss": "cross_entropy", "torchmetrics": ["perplexity", "num_tokens"]},   # pipeline/hg38.yaml:14-17
    "encoder": None, "decoder": None,
    # ---- hg38_hyena.yaml:6-27
    "model": {
        "_name_": "lm", "d_model": 32, "n_layer": 2,
        "d_inner": 128,                                 # eval(4 * 32)
        "vocab_size": 12, "resid_dropout": 0.0, "embed_dropout": 0.1, "fused_mlp": False, "fused_dropout_add_ln": False,
        "checkpoint_mixer": False, "checkpoint_mlp": False, "residual_in_fp32": True, "pad_vocab_size_multiple": 8,
        "layer": {"_name_": "hyena", "emb_dim": 5, "filter_order": 64, "short_filter_order": 3,
                  "l_max": 1026,                        # eval(1024+2)
                  "modulate": True, "w": 10,
                  "lr": 6e-4,                           # ${optimizer.lr}
                  "wd": 0.0, "lr_pos_emb": 0.0},
    },
}
