"""Dimension sweep across every table / lane-mapping boundary of the kernels: psi and grad psi on the device (fp64 and
fp32, through nmpc_eval_batch_*) against the oracle, and three-iteration solves of the throughput, latency and
cooperative kernels against each other, for
  * lanes per step 3 -> 2 -> 1:            N = 20, 21 | 22, 32 | 33, 40, 64
  * register table 4 slots -> 14 -> LDS:   12 | 13, 42 | 43 obstacle rows (N <= 21, fp32)
  * on-chip cooperative table (N > 32):    96 | 97 rows (registers only | + LDS rows), 160, and a row count that no
                                            longer fits LDS (global table)
  * no obstacles at all, one obstacle, obstacle slots that are all padding."""
import numpy as np
import pytest

import dyobav_mpcnwta_warehouse_amd as nm
import oracle
from accuracy_protocol import HOST_THREADS
from dyobav_mpcnwta_warehouse_amd.scenarios import ParamLayout

pytestmark = pytest.mark.gpu

CASES = [  # (N, Ndyn, n_ped, n_hyp)
    (20, 15, 0, 1), (20, 15, 1, 1), (20, 12, 2, 6), (20, 13, 1, 13), (20, 42, 6, 7), (20, 43, 1, 43), (21, 15, 3, 4),
    (22, 15, 3, 5), (32, 20, 4, 5), (33, 24, 4, 6), (40, 96, 8, 12), (40, 97, 1, 97), (40, 160, 8, 20), (40, 400, 20, 20),
    (64, 10, 2, 5), (5, 4, 2, 2),
]


def _cfg(lay, **ov):
    cfg = nm.default_config_struct()
    cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs = lay.N, lay.Nother, lay.Nstc, lay.Ndyn
    for k, v in ov.items():
        setattr(cfg, k, v)
    return cfg


@pytest.mark.parametrize("N,Ndyn,n_ped,n_hyp", CASES)
def test_psi_and_gradient_against_oracle_across_dimensions(N, Ndyn, n_ped, n_hyp):
    lay = ParamLayout(N=N, Ndyn=Ndyn)
    K = 6
    P = nm.scenarios.make_batch(K, lay, seed=100 + N + Ndyn, n_ped=n_ped, n_hyp=n_hyp, ped_mode="oncoming")
    rng = np.random.default_rng(N * 1000 + Ndyn)
    U = np.stack([rng.uniform(-0.3, 1.4, (K, N)), rng.uniform(-0.45, 0.45, (K, N))], axis=2).reshape(K, 2 * N)
    Y = rng.normal(size=(K, 2 * N)) * 2
    C = rng.uniform(1, 200, K)
    pr = oracle.Problem(N, lay.Nother, lay.Nstc, Ndyn)
    want = [oracle.psi(pr, U[i], C[i], Y[i], P[i]) for i in range(K)]
    for reg_table in (0, -1):
        with nm.Handle(_cfg(lay, reg_table=reg_table)) as h:
            for dtype, tp, tg in ((np.float64, 1e-11, 1e-10), (np.float32, 5e-5, 5e-4)):
                r = h.eval(P, U, Y, C, dtype=dtype)
                for i, (v, g) in enumerate(want):
                    assert r["psi"][i] == pytest.approx(v, rel=tp), (reg_table, dtype, i)
                    np.testing.assert_allclose(r["grad"][i], g, rtol=0, atol=tg * max(1.0, np.abs(g).max()))


@pytest.mark.parametrize("N,Ndyn,n_ped,n_hyp", CASES)
def test_short_solves_agree_across_kernels_and_dimensions(N, Ndyn, n_ped, n_hyp):
    lay = ParamLayout(N=N, Ndyn=Ndyn)
    P = nm.scenarios.make_batch(8, lay, seed=200 + N + Ndyn, n_ped=n_ped, n_hyp=n_hyp, ped_mode="oncoming")
    pr = oracle.Problem(N, lay.Nother, lay.Nstc, Ndyn)
    short = dict(max_outer_iterations=1, max_inner_iterations=3, lip_eps_f64=1e-4, lip_delta_f64=1e-4)
    Uo, ro = oracle.solve_batch(pr, oracle.Options(max_outer=1, max_inner=3, lip_delta=1e-4, lip_eps=1e-4), P, nthreads=HOST_THREADS)
    for dtype, tol in ((np.float64, 1e-7), (np.float32, 2e-2)):
        runs = {}
        for name, ov in (("throughput", dict(latency_waves=1, coop_waves=1)), ("latency", dict(latency_waves=3, coop_waves=1)),
                         ("cooperative", dict(latency_waves=1, coop_waves=4)),
                         ("cooperative-lds", dict(latency_waves=1, coop_waves=4, reg_table=-1)),
                         ("automatic", dict())):
            with nm.Handle(_cfg(lay, **short, **ov)) as h:
                runs[name] = h.solve(P.astype(dtype), dtype=dtype)
        for name, r in runs.items():
            assert np.isfinite(r["U"]).all() and set(np.unique(r["status"])) <= {0, 1}, name
            assert np.array_equal(r["iters"][:, 1], ro["inner_iters"]) or dtype == np.float32, name
            du = np.abs(r["U"].astype(np.float64) - Uo).max(axis=1)
            assert np.median(du) < tol and du.max() < 50 * tol, (name, dtype, np.median(du), du.max())
