"""-m gpu test of the multi-GPU path on the one GPU a test box has: a single-rank RCCL process group ("nccl" backend
of torch.distributed on ROCm), the ragged all-gather and the device-resident sharded solve (no host hop). The
world_size-2 logic is covered on CPU by tests/test_sharding_gloo.py; the 1/2/4/8-GPU scaling curve is the driver's."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

import dyobav_mpcnwta_warehouse_amd as nm
from dyobav_mpcnwta_warehouse_amd.sharding import all_gather_ragged, device_solver, shard_bounds, solve_sharded

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl_group():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


def test_ragged_all_gather_over_rccl(rccl_group):
    t = torch.arange(7, dtype=torch.float32, device="cuda").reshape(7, 1) * 2
    g = all_gather_ragged(t, [7])
    assert g.device.type == "cuda" and torch.equal(g, t)
    assert shard_bounds(7, 1, 0) == (0, 7)


def test_device_resident_sharded_solve_equals_plain_solve(rccl_group):
    L = nm.scenarios.ParamLayout()
    P = nm.scenarios.make_batch(96, L, seed=51, ped_mode="passing").astype(np.float32)
    cfg = nm.default_config_struct()
    cfg.max_active_dynobs = 10
    with nm.Handle(cfg) as h:
        want = h.solve(P)
        dP = torch.from_numpy(P).cuda()
        got = solve_sharded(dP, device_solver(h, np.float32))
        torch.cuda.synchronize()
        for k in ("U", "cost", "status", "iters"):
            assert isinstance(got[k], torch.Tensor) and got[k].device.type == "cuda"      # never left the device
            assert np.array_equal(got[k].cpu().numpy(), want[k]), k
        # numpy in / numpy out still works (host driver of the CPU tests)
        got_np = solve_sharded(P, lambda rows: h.solve(rows))
        assert np.array_equal(got_np["U"], want["U"])
