#!/usr/bin/env python3
"""Randomised differential check (by hand on the GPU box for long runs; a reduced run with fixed seeds is part of the
-m gpu suite through tests/test_gpu_fuzz.py):
psi / grad psi / ||F2||^2 through nmpc_eval_batch_* against the fp64 oracle for random problem dimensions, random
numbers of active obstacle rows placed on the robot's path (so that soft AND hard ellipse terms are active), non-zero
fleet robots and boxes, in every evaluation code path: register / LDS / global obstacle table with one wavefront,
cooperative evaluation with 2..4 wavefronts, the on-chip cooperative kernel with and without helper lanes.
    python tests/fuzz_eval.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dyobav_mpcnwta_warehouse_amd as nm   # noqa: E402
import oracle                               # noqa: E402
from dyobav_mpcnwta_warehouse_amd.scenarios import ParamLayout  # noqa: E402


def make_case(rng, axis_aligned=None):
    """`axis_aligned`: every ellipse with angle 0 (what the reference's producer writes, and what selects the AXIS kernel
    variants); None = one case in three."""
    if axis_aligned is None:
        axis_aligned = rng.random() < 0.34
    N = int(rng.choice([rng.integers(3, 65), 20, 21, 22, 32, 33, 40, 42, 43, 64]))
    Nother, Nstc = int(rng.integers(1, 13)), int(rng.integers(1, 15))
    Ndyn = int(rng.choice([rng.integers(1, 30), rng.integers(30, 220), 12, 13, 42, 43, 96, 97, 144, 145, 160]))
    rows = int(rng.choice([0, 1, 2, 3, Ndyn, rng.integers(0, Ndyn + 1)]))
    lay = ParamLayout(N=N, Nother=Nother, Nstc=Nstc, Ndyn=Ndyn)
    B = 4
    P = nm.scenarios.make_batch(B, lay, seed=int(rng.integers(1 << 30)), n_ped=0, n_hyp=1, n_boxes=min(4, Nstc), ped_mode="oncoming")
    s0 = P[:, lay.s0:lay.s0 + 3]
    ref = P[:, lay.rs:lay.rs + 3 * N].reshape(B, N, 3)
    od = np.zeros((B, Ndyn, N + 1, 6))
    slots = rng.permutation(Ndyn)[:rows]
    for b in range(B):
        path = np.r_[s0[b:b + 1, :2], ref[b, :, :2]]
        for j in slots:
            ctr = path[rng.integers(0, N)] + rng.normal(0, 0.3, 2)
            od[b, j, :, 0:2] = ctr + np.arange(N + 1)[:, None] * rng.normal(0, 0.03, 2)
            od[b, j, :, 2:4] = rng.uniform(0.2, 0.9, 2)
            od[b, j, :, 4] = rng.uniform(-1.5, 1.5) if (rng.random() < 0.7 and not axis_aligned) else 0.0
            od[b, j, :, 5] = rng.uniform(0.0, 1.0, N + 1)
    # hypotheses of one pedestrian share their t = 0 snapshot (what the reference's producer writes): in half of the
    # cases the rows are dealt into a few groups with identical (x, y, rx, ry, angle) at t = 0 and their own weights --
    # the kernels merge such rows for the t = 0 terms
    if rows >= 2 and rng.random() < 0.5:
        n_groups = int(rng.integers(1, max(2, rows // 2 + 1)))
        leaders = slots[:n_groups]
        for j in slots[n_groups:]:
            od[:, j, 0, 0:5] = od[:, leaders[int(rng.integers(0, n_groups))], 0, 0:5]
    P[:, lay.od:lay.od + od[0].size] = od.reshape(B, -1)
    # fleet: some robots with non-zero positions near the path
    for j in range(1, Nother):
        if rng.random() < 0.4:
            P[:, lay.c0 + 3 * j:lay.c0 + 3 * j + 2] = s0[:, :2] + rng.normal(0, 1.0, (B, 2))
            P[:, lay.c + 3 * N * j:lay.c + 3 * N * (j + 1)] = (ref + rng.normal(0, 0.5, ref.shape)).reshape(B, -1)
    U = np.stack([rng.uniform(0.3, 1.4, (B, N)), rng.uniform(-0.3, 0.3, (B, N))], axis=2).reshape(B, 2 * N)
    Y = rng.normal(size=(B, 2 * N))
    C = rng.uniform(1, 300, B)
    return lay, rows, P, U, Y, C


def run(cases=100, seed=0, out=print):
    """returns 0 when every evaluation agrees with the oracle"""
    rng = np.random.default_rng(seed)
    worst = {}
    kinks = []
    n_checks = 0
    for ci in range(cases):
        lay, rows, P, U, Y, C = make_case(rng)
        pr = oracle.Problem(lay.N, lay.Nother, lay.Nstc, lay.Ndyn)
        B = P.shape[0]
        modes = [("one-wave", dict(coop_waves=1, reg_table=0)), ("one-wave/lds", dict(coop_waves=1, reg_table=-1)),
                 ("one-wave/general", dict(coop_waves=1, reg_table=0, axis_aligned=-1)),
                 ("one-wave/reg64", dict(coop_waves=1, reg_table=1)),     # fp64: register-table kernel wherever it is offered
                 ("coop4", dict(coop_waves=4, reg_table=0)), ("coop4/lds", dict(coop_waves=4, reg_table=-1)),
                 ("coop%d" % (2 + ci % 2), dict(coop_waves=2 + ci % 2, reg_table=-1))]
        for dtype, tp, tg in ((np.float64, 1e-10, 1e-9), (np.float32, 2e-4, 2e-3)):
            Pd = P.astype(dtype)
            want = []
            for i in range(B):
                u, y, c, p = (a.astype(dtype).astype(np.float64) for a in (U[i], Y[i], C[i:i + 1], Pd[i]))
                v, g = oracle.psi(pr, u, float(c[0]), y, p)
                f2 = np.asarray(oracle.eval_problem(pr, u, p)[2])
                want.append((v, g, float(np.sum(f2 ** 2))))
            for name, ov in modes:
                cfg = nm.default_config_struct()
                cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs = lay.N, lay.Nother, lay.Nstc, lay.Ndyn
                cfg.latency_waves = 1
                for k, v in ov.items():
                    setattr(cfg, k, v)
                with nm.Handle(cfg) as h:
                    r = h.eval(Pd, U, Y, C, dtype=dtype)
                for i, (v, g, f2) in enumerate(want):
                    ep = abs(r["psi"][i] - v) / max(1.0, abs(v))
                    eg = np.abs(r["grad"][i] - g).max() / max(1.0, np.abs(g).max())
                    ef = abs(r["f2sq"][i] - f2) / max(1.0, abs(f2))
                    key = (name, np.dtype(dtype).name)
                    n_checks += 1
                    if not (ep < tp and eg < tg and ef < 10 * tp):
                        # grad psi jumps where a horizon step sits ON the boundary of a hard ellipse (max(0, h) switches
                        # its gradient on): if the oracle's own gradient moves as much under a 1e-6 perturbation of u,
                        # the two precisions merely landed on different sides of such a kink
                        u64, y64, p64 = (a.astype(dtype).astype(np.float64) for a in (U[i], Y[i], Pd[i]))
                        c64 = float(np.asarray(C[i]).astype(dtype))
                        jump = max(np.abs(oracle.psi(pr, u64 * (1 + s), c64, y64, p64)[1] - g).max() for s in (1e-6, -1e-6))
                        if jump / max(1.0, np.abs(g).max()) > 0.3 * eg:
                            kinks.append((ci, name, np.dtype(dtype).name, i))
                            continue
                        out(f"MISMATCH case {ci} N={lay.N} Nother={lay.Nother} Nstc={lay.Nstc} Ndyn={lay.Ndyn} rows={rows} "
                              f"mode={name} dtype={np.dtype(dtype).name} instance {i}: psi {ep:.2e} grad {eg:.2e} f2 {ef:.2e}")
                        return 1
                    w = worst.setdefault(key, [0.0, 0.0, 0.0])
                    w[0], w[1], w[2] = max(w[0], ep), max(w[1], eg), max(w[2], ef)
    out(f"{cases} cases, {n_checks} evaluations checked, {len(kinks)} of them on a gradient kink (skipped: {kinks[:4]}); "
        f"worst relative errors (psi, grad, f2sq):")
    for k, w in sorted(worst.items()):
        out(f"  {k[0]:16s} {k[1]:8s} {w[0]:.2e} {w[1]:.2e} {w[2]:.2e}")
    return 0


if __name__ == "__main__":
    sys.exit(run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 0))
