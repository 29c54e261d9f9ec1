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


# ---- one large MSM split by point range (optional) -----------------------------
def point_range_for_rank(n: int, rank: int, world: int):
    """[lo, hi) of the bases / scalars rank `rank` owns: contiguous, sizes differ by at most one."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def combine_partial_msm(ctx, curve_id: int, local_partial, rank: int, world: int):
    """local_partial: (1, 12) int64 tensor on this rank's device (the MSM over its point range), COMPLETE on torch's current
    stream (synchronise the context that produced it first) -> (1, 12) tensor holding the whole MSM on every rank: one
    all-gather of world x 96 B, then world - 1 group additions on the device (dehalo_point_sum_device).
    The gather, the re-ordering and the zero-fill of the output run on torch's current stream and are waited for before the
    addition is enqueued on the context's own (non-blocking) stream: nothing else orders the two streams."""
    import torch

    parts = all_gather_commitments(local_partial, world, rank, world) if world > 1 else local_partial
    parts = parts.contiguous()
    out = torch.zeros((1, 12), dtype=parts.dtype, device=parts.device)
    torch.cuda.current_stream().synchronize()
    ctx.point_sum_device(curve_id, parts.data_ptr(), parts.shape[0], out.data_ptr(), 0)
    ctx.synchronize()
    return out
