"""Multi-GPU layer: one process per GPU; the independent units of the path
(proofs of a batch, or the committed columns of one phase) are dealt round-robin to ranks and
need no data-path collective.  The one real exchange is the all-gather of the commitment
vector (96 B Jacobian per commitment, a few KB per proof): latency-bound over xGMI, so a
single RCCL all_gather of equal-sized padded blocks, not a ring of per-unit sends."""
from __future__ import annotations

from typing import List


def units_for_rank(units: int, rank: int, world: int) -> List[int]:
    return list(range(rank, units, world))


def all_gather_commitments(local, units: int, rank: int, world: int):
    """local: (len(units_for_rank), 12) int64 tensor on this rank's device -> (units, 12) on
    every rank, in unit order.  backend "nccl" is RCCL on ROCm; "gloo" in the CPU tests."""
    import torch
    import torch.distributed as dist

    per = (units + world - 1) // world
    pad = torch.zeros((per, local.shape[1]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    if world == 1:
        gathered = pad.unsqueeze(0)
    else:
        out = torch.empty((world, per, local.shape[1]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out.view(world * per, local.shape[1]), pad)
        gathered = out
    # unit u lives at [u % world][u // world]
    full = gathered.permute(1, 0, 2).reshape(per * world, local.shape[1])[:units]
    return full.contiguous()


def max_over_ranks(value: float) -> float:
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
