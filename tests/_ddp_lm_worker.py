"""Worker of tests/test_distributed_cpu.py::test_ddp_lm_world_size_2 (launched by torch.distributed.run).
The REAL model of bench.py's N > 1 leg -- hyena_dna_amd.lm.HyenaDNALM wrapped in DistributedDataParallel with the reference
trainer's settings (train.py:611-620: find_unused_parameters=False, gradient_as_bucket_view=True) -- on gloo, kernels under
tests/hipemu: every rank runs forward + loss + backward on ITS sequences; the all-reduced (averaged) gradients of every
parameter must equal those of one process over the whole batch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel as DDP


def main():
    from hyena_dna_amd import _lib
    from tests.hipemu.emu_backend import EmuBackend
    _lib._backend = EmuBackend()
    from hyena_dna_amd.lm import HyenaDNALM

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)                                   # identical initial weights on every rank
    D, L, B = 64, 96, 4
    layer = dict(l_max=L + 2, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10, lr=6e-4, wd=0.0,
                 lr_pos_emb=0.0)
    model = HyenaDNALM(d_model=D, n_layer=2, d_inner=4 * D, vocab_size=12, layer=layer, resid_dropout=0.0, embed_dropout=0.0,
                       pad_vocab_size_multiple=8, fused_dropout_add_ln=True, residual_in_fp32=True)
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(7, 11, (B, L), generator=g)
    tgt = torch.roll(ids, -1, 1)

    def loss_of(net, sl):
        logits = net(ids[sl])[0].logits
        return torch.nn.functional.cross_entropy(logits.float().reshape(-1, logits.shape[-1]), tgt[sl].reshape(-1))

    ref = None
    if rank == 0:                                          # one process over the whole batch: mean over B sequences
        loss_of(model, slice(0, B)).backward()
        ref = {n: p.grad.clone() for n, p in model.named_parameters()}
        model.zero_grad(set_to_none=True)
    ddp = DDP(model, find_unused_parameters=False, gradient_as_bucket_view=True)
    shard = slice(rank * B // world, (rank + 1) * B // world)
    loss_of(ddp, shard).backward()                         # per-rank mean over B / world sequences; DDP averages over ranks
    if rank == 0:
        worst = 0.0
        for n, p in model.named_parameters():
            assert p.grad is not None, n
            err = ((p.grad - ref[n]).norm() / ref[n].norm().clamp_min(1e-30)).item()
            worst = max(worst, err)
            assert err < 2e-5, (n, err)
        print(f"DDP_LM_OK world={world} params={len(ref)} worst_rel={worst:.2e}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
