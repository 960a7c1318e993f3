"""Data parallelism over the GPUs of one node: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The path shards by minibatch rows: every per-row computation is independent, parameters are replicated, and
there is ONE real exchange step per optimiser group — the gradient mean.  Because each group's gradients live in one
contiguous arena (arena.py) that exchange is a single all-reduce per group (109 MB score group, 2-5 MB the others), with
no bucketing or flatten/unflatten copies.  The three batch-coupled pieces get global semantics with tiny collectives:
  * reward normaliser: all-gather the B rewards of each rank, update the fp64 statistics from the global batch;
  * time-importance EMA: all-gather (t, per-sample loss) and replay the sequential EMA in global (rank-major) order;
  * lambda-returns (minibatch order is time, window 5): a 5-row halo — every rank appends the first n rows of the next rank's
    (next_latents, normalised rewards, dones) to its own, so the windows of its last rows see the neighbours they would see in
    one global batch (`halo_from_next_rank`); the last rank gets none.  N ranks x B rows then equal one N*B-row step
    (tests/test_hip_dp_equiv.py).
  * KL annealing weight exp(-5 mean t): mean over the all-gathered t (the same gather feeds the time-importance replay).
  * MINE statistics stay per rank: gradient-free, they enter only the logged EFE value and the estimator's running mean.
Nothing here is GPU-specific: the same code runs on CPU tensors under the gloo backend (tests/test_parallel_gloo.py).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


def is_dp() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def allreduce_mean_(flat: torch.Tensor, async_op: bool = False):
    """In-place mean over ranks of one flat gradient arena.  AVG on RCCL; SUM + scale on backends without AVG (gloo)."""
    if not is_dp():
        return None
    if dist.get_backend() == "nccl":
        return dist.all_reduce(flat, op=dist.ReduceOp.AVG, async_op=async_op)
    h = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=False)
    flat.div_(dist.get_world_size())
    return h if async_op else None


def agree_min(values, device=None) -> List[int]:
    """Element-wise MIN over ranks of a short list of host integers (one tiny all-reduce).  This is how rank-local DECISIONS that gate
    collectives are made collective: every rank contributes what it would do alone and all of them act on the minimum, so the sequence of
    collectives a step issues can never differ between ranks (a MAX is the MIN of the negated value).  RCCL needs a device tensor
    (`device`); gloo takes a host tensor.  Without a process group the values come back unchanged."""
    vals = [int(v) for v in values]
    if not is_dp():
        return vals
    on_gpu = dist.get_backend() == "nccl"
    t = torch.tensor(vals, dtype=torch.int64, device=device if on_gpu else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return [int(v) for v in t.tolist()]


def agree_max_float(value: float, device=None) -> float:
    """MAX over ranks of one host float (step-time votes of the start-up schedule check)."""
    if not is_dp():
        return float(value)
    on_gpu = dist.get_backend() == "nccl"
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if on_gpu else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allgather_cat(x: torch.Tensor) -> torch.Tensor:
    """Concatenate a same-shaped tensor from every rank along dim 0, rank-major."""
    if not is_dp():
        return x
    out = torch.empty((dist.get_world_size() * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x.contiguous())
    return out


def broadcast_arena_(flat: torch.Tensor, src: int = 0):
    """Make every replica start from rank `src`'s parameters."""
    if is_dp():
        dist.broadcast(flat, src=src)


def shard_indices(global_indices, rank_: Optional[int] = None, world_: Optional[int] = None):
    """Disjoint, contiguous slice of one global index draw (exact 1-GPU == N-GPU tests): rank r takes rows
    [r*B, (r+1)*B) of the global batch."""
    r = rank() if rank_ is None else rank_
    w = world() if world_ is None else world_
    n = len(global_indices)
    if n % w:
        raise ValueError(f"global batch {n} is not divisible by world size {w}")
    b = n // w
    return global_indices[r * b:(r + 1) * b]


def halo_from_next_rank(*tensors: torch.Tensor, rows: int = 5):
    """For each [B, ...] tensor: this rank's rows followed by the first `rows` rows of rank+1's (nothing appended on the last rank).
    One small all-gather per tensor group: the heads of all ranks are gathered, each rank keeps its successor's."""
    if not is_dp():
        return tensors
    w, r = dist.get_world_size(), dist.get_rank()
    out = []
    for x in tensors:
        n = min(rows, x.shape[0])
        heads = allgather_cat(x[:n].contiguous().reshape(1, n, -1))          # [W, n, F]
        if r + 1 < w:
            out.append(torch.cat([x, heads[r + 1].reshape((n,) + tuple(x.shape[1:]))], dim=0).contiguous())
        else:
            out.append(x)
    return tuple(out)
