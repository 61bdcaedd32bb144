"""Worker of tests/test_distributed_cpu.py::test_ddp_operator_world_size_2 (launched by torch.distributed.run).
Each rank wraps the SAME HyenaOperator (kernels under tests/hipemu, gloo) in DistributedDataParallel, runs forward +
backward on ITS shard of a fixed batch, and rank 0 compares the all-reduced gradients with a single-process run over
the whole batch -- the reference's multi-GPU scheme (train.py:611-620: DDP over the batch axis, nothing else)."""
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
    from hyena_dna_amd.hyena import HyenaOperator

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)                                   # identical initial weights on every rank
    D, L, B = 64, 96, 4
    op = HyenaOperator(d_model=D, l_max=L + 2, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10,
                       lr=6e-4, wd=0.0, lr_pos_emb=0.0)
    g = torch.Generator().manual_seed(1)
    u = torch.randn(B, L, D, generator=g)
    dy = torch.randn(B, L, D, generator=g)
    ref = None
    if rank == 0:                                          # single-process truth over the whole batch (mean over ranks = /world)
        op(u).backward(dy / world)
        ref = {n: p.grad.clone() for n, p in op.named_parameters()}
        op.zero_grad(set_to_none=True)
    ddp = DDP(op)
    shard = slice(rank * B // world, (rank + 1) * B // world)
    ddp(u[shard]).backward(dy[shard])                      # DDP averages the per-rank gradients
    if rank == 0:
        worst = 0.0
        for n, p in op.named_parameters():
            assert p.grad is not None, n
            err = ((p.grad - ref[n]).norm() / ref[n].norm().clamp_min(1e-30)).item()
            worst = max(worst, err)
            assert err < 2e-5, (n, err)
        print(f"DDP_OK world={world} params={len(ref)} worst_rel={worst:.2e}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
