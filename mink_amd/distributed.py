"""Batch sharding across the GPUs of one node: one process per GPU with an RCCL gather of v (bench.py --gpus N,
solve_ik_sharded), or ONE process that drives every device (ShardedProblem: SURVEY §8(e)'s second host model).

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


class ShardedProblem:
    """One `NativeProblem` per device and a HOST batch split into contiguous row blocks (`shard_bounds`): the single-process
    host model of SURVEY §8(e).  `devices` lists the device of every shard — repeats allowed: N handles on ONE device is how
    this path is exercised on a 1-GPU box.  The shards' host-pointer calls are issued from N threads: a libminkhip call
    releases the GIL and is synchronous per handle (its own staging buffers and streams), so the devices copy and compute
    concurrently; results land in row blocks of one output.  Same `solve` signature and return values as
    `NativeProblem.solve` for numpy inputs (taps, fused steps, plugin rows included).  No collective: the instances are
    independent, and here the rows come home over PCIe anyway."""

    def __init__(self, model, devices, max_batch: int, **problem_kwargs):
        from concurrent.futures import ThreadPoolExecutor

        from . import _native as nat

        self.devices = [int(d) for d in devices]
        if not self.devices:
            raise ValueError("ShardedProblem needs at least one device")
        n = len(self.devices)
        self.max_batch = int(max_batch)
        shard_max = -(-self.max_batch // n)
        self.models = [nat.NativeModel(model, d) for d in self.devices]            # one handle per shard, also on a shared device
        self.shards = [nat.NativeProblem(nm, max_batch=shard_max, **problem_kwargs) for nm in self.models]
        self._pool = ThreadPoolExecutor(max_workers=n)
        p0 = self.shards[0]
        self.n_frame, self.n_posture, self.n_com, self.n_rows, self.n_pairs = p0.n_frame, p0.n_posture, p0.n_com, p0.n_rows, p0.n_pairs
        self.n_dense_rows, self.n_dense_limit_rows, self.dense_limit_box = p0.n_dense_rows, p0.n_dense_limit_rows, p0.dense_limit_box

    def close(self):
        for p in getattr(self, "shards", []):
            p.close()
        for m in getattr(self, "models", []):
            m.close()
        self.shards, self.models = [], []
        if getattr(self, "_pool", None) is not None:
            self._pool.shutdown(wait=False)
            self._pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_kernel(self) -> str:
        return self.shards[0].last_kernel()

    def launch_info(self, B: int):
        return self.shards[0].launch_info(-(-int(B) // len(self.shards)))

    def solve(self, q, frame_targets=None, posture_target=None, com_target=None, dt: float = 1e-2, damping: float = 1e-12,
              **kw):
        q = np.ascontiguousarray(q, dtype=np.float64)
        B, n = int(q.shape[0]), len(self.shards)
        if B > self.max_batch:
            from ._native import MinkHipError
            raise MinkHipError(f"B={B} exceeds max_batch={self.max_batch} of this problem")
        for bad in ("out", "status_out", "q_out"):
            if kw.get(bad) is not None:
                raise ValueError(f"ShardedProblem.solve allocates its outputs ('{bad}' is not supported)")
        bounds = [shard_bounds(B, n, r) for r in range(n)]

        # posture / CoM targets: batched iff they carry the extra leading axis
        def tgt(x, per):
            if x is None:
                return [None] * n
            x = np.asarray(x)
            return [x[lo:hi] if x.ndim == per + 1 else x for lo, hi in bounds]

        pts, cts = tgt(posture_target, 2), tgt(com_target, 2)
        dense = kw.pop("dense", None)
        jobs = []
        for r, (lo, hi) in enumerate(bounds):
            if hi == lo:
                jobs.append(None)
                continue
            d_r = None if dense is None else {k: np.ascontiguousarray(np.asarray(v)[lo:hi]) for k, v in dense.items()}
            jobs.append(self._pool.submit(self.shards[r].solve, q[lo:hi], None if frame_targets is None else np.asarray(frame_targets)[lo:hi],
                                          pts[r], cts[r], dt, damping, dense=d_r, **kw))
        parts = [j.result() for j in jobs if j is not None]

        def join(items):
            if items[0] is None:
                return None
            if isinstance(items[0], dict):
                return {k: join([it[k] for it in items]) for k in items[0]}
            return np.concatenate(items, axis=0)

        return tuple(join([p[i] for p in parts]) for i in range(len(parts[0])))
