"""Batch sharding across the GPUs of one node: one process per GPU, RCCL gather of v.

The IK instances are independent (no cross-instance term anywhere in mink/solve_ik.py:13-40), so
rank r owns the contiguous rows [r·B/N, (r+1)·B/N) and the model constants are replicated at handle
creation.  The single optional collective is a gather of `v` (and `status`) to one rank.  Works with
any torch.distributed backend: `nccl` (= RCCL over xGMI) on GPUs, `gloo` in the CPU tests.
"""

from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


def shard_bounds(n: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced row range of `rank` (first n % world_size ranks get one extra row)."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of size {world_size}")
    base, extra = divmod(n, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard(array, world_size: int, rank: int):
    lo, hi = shard_bounds(array.shape[0], world_size, rank)
    return array[lo:hi]


def gather_rows(local, total_rows: int, dst: int = 0, group=None, out=None):
    """Gather row-sharded tensors (shards from `shard_bounds`) to rank `dst`.

    Returns the (total_rows, ...) tensor on `dst`, None elsewhere.  With even shards the collective writes
    straight into row blocks of the result (`out`, if given, is reused: no allocation, no concatenation);
    uneven shards are padded to the largest shard and trimmed afterwards (RCCL gather wants equal counts).
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_bounds(total_rows, world, r) for r in range(world)]
    max_rows = max(hi - lo for lo, hi in sizes)
    even = all(hi - lo == max_rows for lo, hi in sizes)
    if even:
        bufs = None
        if rank == dst:
            if out is None:
                out = torch.empty((total_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            bufs = [out[lo:hi] for lo, hi in sizes]           # contiguous row blocks of the result
        dist.gather(local.contiguous(), bufs, dst=dst, group=group)
        return out if rank == dst else None
    pad = local
    if local.shape[0] < max_rows:
        pad = torch.zeros((max_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad.contiguous(), bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], dim=0)


def solve_ik_sharded(solve_local, q, frame_targets, total_rows: Optional[int] = None, dst: int = 0,
                     gather: bool = True, group=None):
    """Run `solve_local(q_shard, targets_shard) -> (v, status)` on this rank's shard of a batch that is
    already sharded (pass the local rows) and gather v/status to `dst` (optional)."""
    import torch.distributed as dist

    v, status = solve_local(q, frame_targets)
    if not gather:
        return v, status
    world = dist.get_world_size(group)
    if total_rows is None:
        total_rows = int(q.shape[0]) * world
    return gather_rows(v, total_rows, dst, group), gather_rows(status, total_rows, dst, group)
