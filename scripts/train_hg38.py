#!/usr/bin/env python
"""Train the reference's hg38 HyenaDNA experiment without Lightning / Hydra (hyena_dna_amd.runner):

    python scripts/train_hg38.py [--configs <reference>/configs | tests/golden/hg38_hyena_composed.json] [--steps 50]
        [--synthetic-genome DIR] [--graphed] [key=value ...]

`key=value` are Hydra-style overrides of configs/experiment/hg38/hg38_hyena.yaml, e.g.
    dataset.max_length=32768 dataset.batch_size=8 model.d_model=256 model.n_layer=8 model.fused_dropout_add_ln=true
`--synthetic-genome DIR` writes a small FASTA + BED there and points dataset.fasta_file / bed_file at them (there is no
hg38.ml.fa on the GPU boxes).  Launch under torch.distributed.run for several GPUs (one process per GPU, DDP over RCCL).
Prints one JSON line with the losses at the end; exit code 1 if the loss did not fall."""
import argparse
import json
import os
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # before the HIP runtime starts: see hyena_dna_amd/__init__.py
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ref = os.path.join(os.environ.get("HYENA_REFERENCE", "/root/reference"), "configs")
    ap.add_argument("--configs", default=ref if os.path.isdir(ref) else os.path.join(ROOT, "tests", "golden", "hg38_hyena_composed.json"))
    ap.add_argument("--experiment", default="hg38/hg38_hyena")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--synthetic-genome", default=None)
    ap.add_argument("--graphed", action="store_true", help="capture the whole step into one hipGraph (lm.GraphedTrainStep); needs trainer.precision=bf16 or 32 -- the shipped config's "
                         "precision 16 is fp16 + a host-side loss scaler, which a captured step cannot skip updates for")
    ap.add_argument("--emu", action="store_true", help="TEST ONLY: kernels under tests/hipemu on the CPU")
    ap.add_argument("overrides", nargs="*")
    args = ap.parse_args()

    from hyena_dna_amd import _lib, runner
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.emu:
        from tests.hipemu.emu_backend import EmuBackend
        _lib._backend = EmuBackend()
        dev = torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "needs a ROCm device (there is no CPU path)"
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        runner.set_affinity(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.distributed.init_process_group("gloo" if args.emu else "nccl", rank=rank, world_size=world)
    overrides = list(args.overrides)
    if args.synthetic_genome:
        max_len = next((int(o.split("=")[1]) for o in overrides if o.startswith("dataset.max_length=")), 1024)
        if rank == 0:
            runner.make_synthetic_genome(args.synthetic_genome, chr_len=max(400_000, 4 * max_len), interval_len=max_len)
        if world > 1:
            torch.distributed.barrier()
        overrides += [f"dataset.fasta_file={os.path.join(args.synthetic_genome, 'synthetic.fa')}",
                      f"dataset.bed_file={os.path.join(args.synthetic_genome, 'synthetic.bed')}"]
    cfg = runner.compose(args.configs, args.experiment, overrides)
    log = print if rank == 0 else (lambda *a, **k: None)
    losses = runner.train(cfg, args.steps, dev, graphed=args.graphed, log=log)
    if rank == 0:
        n = max(1, len(losses) // 5)
        first, last = sum(losses[:n]) / n, sum(losses[-n:]) / n
        print(json.dumps({"steps": len(losses), "loss_first": first, "loss_last": last, "fell": last < first, "losses": losses,
                          "gpu_mem": cfg["train"].get("gpu_mem"), "world": world, "graphed": args.graphed}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0 if (rank != 0 or last < first) else 1


if __name__ == "__main__":
    sys.exit(main())
