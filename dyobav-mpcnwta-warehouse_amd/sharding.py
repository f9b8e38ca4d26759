"""Multi-GPU: the batch of independent MPC problems is split into contiguous shards, one process per GPU
(``torch.distributed``; backend "nccl" is RCCL on ROCm, "gloo" in the CPU tests). The solve needs no
communication; the only collective is the gather of the results: ``(B/G) x (2N)`` controls plus
``(B/G) x 4`` scalars per rank -- 11.5 MB per GPU at B = 524288, far below one xGMI link-second.

The reference has no counterpart (its evaluation loop ``main_base.py:448-464`` is sequential).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import numpy as np


def shard_bounds(B: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced partition: ranks < B % G get one extra instance. Returns [lo, hi)."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside [0, {world_size})")
    base, extra = divmod(B, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_ragged(local, counts, group=None):
    """All-gather of per-rank tensors whose leading dimension differs by at most one (``shard_bounds``): pad to the
    largest shard, one ``all_gather_into_tensor``, strip the padding."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    mx = max(counts)
    pad = local
    if local.shape[0] < mx:
        pad = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))], dim=0)
    out = local.new_empty((world * mx,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    return torch.cat([out[r * mx:r * mx + counts[r]] for r in range(world)], dim=0)


def solve_sharded(P_full, solve_local: Callable, device: str = "cuda", group=None) -> Dict[str, object]:
    """Every rank solves rows ``shard_bounds(B, G, rank)`` of ``P_full`` with ``solve_local`` (e.g. ``Handle.solve`` or
    a device-side wrapper of ``Handle.solve_raw``) and receives the gathered ``U``, ``cost``, ``status``, ``iters`` of
    the whole batch.

    ``P_full`` must be the same array on every rank (deterministic generators make that free); only the local rows
    are touched. It may be a numpy array or a torch tensor; ``solve_local`` gets the local rows in the same kind.
    Results that ``solve_local`` returns as torch tensors on ``device`` are gathered **in place on the device** (RCCL
    over xGMI; no host hop -- the 92 MB of BASELINE configs[3] never leave HBM) and returned as device tensors; numpy
    results are moved to ``device`` for the collective and returned as numpy arrays.
    """
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    B = P_full.shape[0]
    lo, hi = shard_bounds(B, world, rank)
    counts = [shard_bounds(B, world, r)[1] - shard_bounds(B, world, r)[0] for r in range(world)]
    local = P_full[lo:hi]
    res = solve_local(local.contiguous() if hasattr(local, "contiguous") else np.ascontiguousarray(local))
    out = {}
    for key in ("U", "cost", "status", "iters"):
        v = res[key]
        if isinstance(v, torch.Tensor):
            out[key] = all_gather_ragged(v if v.device.type == torch.device(device).type else v.to(device), counts, group)
        else:
            t = torch.from_numpy(np.ascontiguousarray(v)).to(device)
            out[key] = all_gather_ragged(t, counts, group).cpu().numpy()
    return out


def device_solver(handle, dtype=np.float32):
    """``solve_local`` for ``solve_sharded`` that keeps everything in HBM: takes the local rows as a device tensor and
    returns device tensors (``Handle.solve_raw`` on torch's current stream)."""
    import torch
    tdt = torch.float32 if np.dtype(dtype) == np.float32 else torch.float64

    def run(P_local):
        B, n = P_local.shape[0], handle.n
        dev = P_local.device
        U = torch.empty(B, n, dtype=tdt, device=dev)
        cost = torch.empty(B, dtype=tdt, device=dev)
        status = torch.empty(B, dtype=torch.int32, device=dev)
        iters = torch.empty(B, 2, dtype=torch.int32, device=dev)
        handle.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        handle.solve_raw(dtype, P_local.to(tdt).contiguous(), B, U, cost, status, iters, sync=False)
        return dict(U=U, cost=cost, status=status, iters=iters)

    return run
