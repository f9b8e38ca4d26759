"""Scheduling never changes results (round 6, late): the automatic plan of a batch -- dispatch order from one evaluation or from a
pilot launch, resumable solve, tail hand-off, wavefronts per instance -- against the same kernel family run plainly (one launch
in index order, no hand-off), on random dimensions, capacity hints, batch sizes and scenario families: every result array
identical, bit for bit."""
import numpy as np
import pytest

import dyobav_mpcnwta_warehouse_amd as nm

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(24))
def test_automatic_plan_against_one_plain_launch(seed):
    rng = np.random.default_rng(1000 + seed)
    n_ped, n_hyp = [(2, 3), (2, 5), (3, 5), (3, 6), (4, 6), (4, 10)][seed % 6]
    ndyn = n_ped * n_hyp + int(rng.integers(0, 3))
    lay = nm.scenarios.ParamLayout(20, 10, 10, ndyn)
    B = [400, 700, 1000, 1500, 2600, 5000, 9000, 14000, 20000, 30000, 1300, 3500][(seed * 7) % 12]
    fam = ["passing", "toward_robot", "oncoming"][int(rng.integers(0, 3))]
    P = nm.scenarios.make_batch_chunked(B, lay, seed=int(rng.integers(1, 1 << 20)), n_ped=n_ped, n_hyp=n_hyp, ped_mode=fam, dtype=np.float32)
    hint = int(rng.choice([0, n_ped * n_hyp]))
    res, infos = {}, {}
    for name, ov in (("automatic", {}), ("plain", dict(staged=-1, tail_latency=-1))):
        cfg = nm.default_config_struct()
        cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs = lay.N, lay.Nother, lay.Nstc, lay.Ndyn
        cfg.max_active_dynobs = hint
        cfg.max_inner_iterations, cfg.max_outer_iterations = 120, 6      # (short solves: the suite's time)
        for k, v in ov.items():
            setattr(cfg, k, v)
        with nm.Handle(cfg) as h:
            res[name] = h.solve(P)
            infos[name] = h.last_launch_info()
    print("plan:", B, ndyn, hint, fam, infos["automatic"])
    assert infos["automatic"]["family"] == infos["plain"]["family"], infos
    for k in ("U", "y", "cost", "status", "iters"):
        assert np.array_equal(res["automatic"][k], res["plain"][k], equal_nan=True), (k, B, ndyn, hint, fam, infos)
    assert np.array_equal(res["automatic"]["info"][:, :6], res["plain"]["info"][:, :6], equal_nan=True), (B, ndyn, hint, fam, infos)
