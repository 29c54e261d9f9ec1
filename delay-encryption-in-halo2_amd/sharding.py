"""Multi-GPU layer: one process per GPU; the independent units of the path
(proofs of a batch, or the committed columns of one phase) are dealt round-robin to ranks and
need no data-path collective.  The one real exchange is the all-gather of the commitment
vector (96 B Jacobian per commitment, a few KB per proof): latency-bound over xGMI, so a
single RCCL all_gather of equal-sized padded blocks, not a ring of per-unit sends."""
from __future__ import annotations

from typing import List


def units_for_rank(units: int, rank: int, world: int) -> List[int]:
    return list(range(rank, units, world))


def all_gather_commitments(local, units: int, rank: int, world: int, force_collective: bool = False):
    """local: (len(units_for_rank), 12) int64 tensor on this rank's device -> (units, 12) on
    every rank, in unit order.  backend "nccl" is RCCL on ROCm; "gloo" in the CPU tests.
    At world 1 there is nothing to exchange and no process group is needed; `force_collective` runs the collective all the same (a one-rank group
    must be initialised): the `-m gpu` test that pushes HBM tensors through RCCL's all_gather_into_tensor on a one-GPU box."""
    import torch
    import torch.distributed as dist

    per = (units + world - 1) // world
    pad = torch.zeros((per, local.shape[1]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    if world == 1 and not force_collective:
        gathered = pad.unsqueeze(0)
    else:
        out = torch.empty((world, per, local.shape[1]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out.view(world * per, local.shape[1]), pad)
        gathered = out
    # unit u lives at [u % world][u // world]
    full = gathered.permute(1, 0, 2).reshape(per * world, local.shape[1])[:units]
    return full.contiguous()


def gather_proof_commitments(local_blobs, total: int, rank: int, world: int, device="cpu", force_collective: bool = False):
    """Batch proving (BASELINE configs[4]: proof p -> rank p mod world): every rank hands in the commitment bytes of the proofs it
    made (prover.proof_commitments, equal lengths), in the order of units_for_rank, and receives all `total` blobs in proof
    order -- ONE all-gather of total x len bytes (64 proofs x 31 x 32 B = 62 KB: latency-bound over xGMI)."""
    import torch

    mine = units_for_rank(total, rank, world)
    if len(local_blobs) != len(mine):
        raise ValueError("rank %d made %d proofs, its share is %d" % (rank, len(local_blobs), len(mine)))
    width = len(local_blobs[0]) if local_blobs else 0
    if world > 1 or force_collective:                         # ranks without a proof still need the common width
        import torch.distributed as dist
        w = torch.tensor([width], dtype=torch.int64, device=device)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        width = int(w.item())
    if any(len(b) != width for b in local_blobs) or width % 8:
        raise ValueError("commitment blobs must have one common length, a multiple of 8 bytes")
    local = torch.zeros((len(mine), width // 8), dtype=torch.int64)
    for j, b in enumerate(local_blobs):
        local[j] = torch.frombuffer(bytearray(b), dtype=torch.int64)
    full = all_gather_commitments(local.to(device), total, rank, world, force_collective).cpu()
    return [full[u].numpy().tobytes() for u in range(total)]


def max_over_ranks(value: float) -> float:
    import torch
    import torch.distributed as dist

    if not dist.is_initialized():
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


def combine_partial_msm(ctx, curve_id: int, local_partial, rank: int, world: int, force_collective: bool = False):
    """local_partial: (1, 12) int64 tensor on this rank's device (the MSM over its point range), COMPLETE on torch's current
    stream (synchronise the context that produced it first) -> (1, 12) tensor holding the whole MSM on every rank: one
    all-gather of world x 96 B, then world - 1 group additions on the device (dehalo_point_sum_device).
    The gather, the re-ordering and the zero-fill of the output run on torch's current stream and are waited for before the
    addition is enqueued on the context's own (non-blocking) stream: nothing else orders the two streams."""
    import torch

    parts = all_gather_commitments(local_partial, world, rank, world, force_collective) if world > 1 or force_collective else local_partial
    parts = parts.contiguous()
    out = torch.zeros((1, 12), dtype=parts.dtype, device=parts.device)
    torch.cuda.current_stream().synchronize()
    ctx.point_sum_device(curve_id, parts.data_ptr(), parts.shape[0], out.data_ptr(), 0)
    ctx.synchronize()
    return out


# ---- one proof sharded by commitment columns (SURVEY.md 8e "single-proof mode"; dehalo_prover_set_shard) -------------
def column_range_for_rank(count: int, rank: int, world: int):
    """[first, first + num) of the `count` columns of a commitment phase whose MSMs rank `rank` runs (the library's own rule)."""
    first = count * rank // world
    return first, count * (rank + 1) // world - first


def gather_points(points, first, num, rank: int, world: int, device="cpu"):
    """The exchange dehalo_prover_set_shard's callback has to make: `points` is a (count, 8) int64 / uint64 array of affine points of which
    rows [first[rank], first[rank] + num[rank]) are this rank's; on return every row holds its owner's point.  ONE all_gather of equal padded
    blocks (at most ceil(count / world) points a rank: a few hundred bytes -- latency-bound over xGMI)."""
    import numpy as np
    import torch
    import torch.distributed as dist

    per = max(int(x) for x in num)
    pad = torch.zeros((max(per, 1), 8), dtype=torch.int64)
    mine = np.asarray(points[int(first[rank]):int(first[rank]) + int(num[rank])]).view(np.int64)
    pad[: mine.shape[0]] = torch.from_numpy(np.ascontiguousarray(mine))
    out = torch.empty((world * pad.shape[0], 8), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, pad.to(device))
    out = out.cpu().numpy().reshape(world, pad.shape[0], 8)
    for r in range(world):
        if r != rank and num[r]:
            points[int(first[r]):int(first[r]) + int(num[r])] = out[r, : int(num[r])].view(points.dtype)
    return points
