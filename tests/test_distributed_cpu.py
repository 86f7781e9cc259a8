"""N>1 path on CPU: world_size-2 gloo processes exercise the sharding + gather logic used by
bench.py / mink_amd.distributed (the solve itself is replaced by a deterministic stand-in —
the HIP kernel needs a GPU)."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mink_amd.distributed import gather_rows, shard, shard_bounds, solve_ik_sharded


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 64, 65536, 524288 + 3):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)
        q_all = torch.from_numpy(rng.normal(size=(total, 5)))
        tg_all = torch.from_numpy(rng.normal(size=(total, 2, 7)))

        def fake_solve(q, tg):                       # stand-in with per-row results
            v = q * 2.0 + tg[:, 0, :5]
            return v, (q[:, 0] > 0).to(torch.int32)

        v, st = solve_ik_sharded(fake_solve, shard(q_all, world, rank), shard(tg_all, world, rank),
                                 total_rows=total, dst=0)
        if rank == 0:
            v_ref, st_ref = fake_solve(q_all, tg_all)
            ret["ok"] = bool(torch.equal(v, v_ref) and torch.equal(st, st_ref))
        else:
            assert v is None and st is None
        # barrier + MAX-reduce of a timing, as bench.py does
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == world
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [64, 37])       # even and ragged shards
def test_world_size_2_gloo_gather(total):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), total, ret), nprocs=2, join=True)
    assert ret.get("ok") is True
