"""world_size-2 gloo test of the N>1 path: shard bounds, ragged all-gather and the sharded-solve driver with a
deterministic stand-in for the per-rank solve (the real per-rank solve is the single-GPU path of the gpu tests)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dyobav_mpcnwta_warehouse_amd.sharding import all_gather_ragged, shard_bounds, solve_sharded


def test_shard_bounds_partition():
    for B in (0, 1, 7, 1024, 1025, 524288):
        for G in (1, 2, 3, 8):
            cuts = [shard_bounds(B, G, r) for r in range(G)]
            assert cuts[0][0] == 0 and cuts[-1][1] == B
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(G - 1))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _fake_solve(P):
    """Deterministic function of the rows only (instances are independent)."""
    U = np.cumsum(P[:, :4], axis=1).astype(np.float32)
    return dict(U=U, cost=P.sum(axis=1).astype(np.float32), status=(P[:, 0] > 0).astype(np.int32),
                iters=np.stack([P[:, 1], P[:, 2]], axis=1).astype(np.int32))


def _worker(rank, world, port, B, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P = np.random.default_rng(0).normal(size=(B, 6)) * 10
        got = solve_sharded(P, _fake_solve, device="cpu")
        want = _fake_solve(P)
        ok = all(np.array_equal(got[k], want[k]) for k in want)
        lo, hi = shard_bounds(B, world, rank)
        t = torch.arange(lo, hi, dtype=torch.float32).reshape(-1, 1)
        counts = [shard_bounds(B, world, r)[1] - shard_bounds(B, world, r)[0] for r in range(world)]
        g = all_gather_ragged(t, counts)
        ok = ok and torch.equal(g[:, 0], torch.arange(B, dtype=torch.float32))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [10, 11])
def test_sharded_solve_two_ranks_gloo(B):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(2))
    assert res == {0: True, 1: True}
