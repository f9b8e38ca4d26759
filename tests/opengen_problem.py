#!/usr/bin/env python3
"""Genuine OpEn solver for the same problem -- BENCH / TEST INFRASTRUCTURE, used only where the box has OpEn's
toolchain (cargo + opengen + casadi). `bench.py` probes for it (`cpu_baseline.open_probe`); neither the authoring
container nor the GPU boxes of rounds 1-2 had it, so THIS FILE HAS NEVER RUN: it records how the open questions of
SURVEY.md 8(c) (AKKT residual form Q2, multiplier carry-over Q1, reported cost Q4) get closed the day a box has OpEn.

The problem is written here from SURVEY.md 8(a) (rows A1-A8), not copied from the reference's mpc_builder.py: same
parameter layout (u_m1, s_0, s_N, q, r_s, r_v, c_0, c, o_s, o_d, q_stc, q_dyn), same cost terms, F1 = accelerations in
C = [amin, amax]^N x [-wamax, wamax]^N (augmented Lagrangian), F2 = per-obstacle hard indicators + polygon scalar
(penalty), U = velocity box; solver configuration of mpc_builder.py:187-195 (initial penalty 10, max_duration).
"""
from __future__ import annotations

import os
import sys
import tempfile
import time

import numpy as np


def build_problem(N=20, Nother=10, Nstc=10, Ndyn=15, ts=0.2, vmin=-0.5, vmax=1.5, wmax=0.5, amin=-1.0, amax=1.0,
                  wamax=3.0, width=0.5, vm=0.2, sm=0.2):
    import casadi as cs
    import opengen as og
    nu, ns = 2, 3
    u = cs.SX.sym("u", nu * N)
    npar = 18 + 6 * N + 3 * Nother * (N + 1) + 12 * Nstc + 6 * Ndyn * (N + 1)
    p = cs.SX.sym("p", npar)
    o = 0

    def take(n):
        nonlocal o
        v = p[o:o + n]
        o += n
        return v
    um1, s0, sN, q = take(2), take(3), take(3), take(10)
    rs, rv = take(3 * N), take(N)
    c0, c = take(3 * Nother), take(3 * N * Nother)
    os_, od = take(12 * Nstc), take(6 * Ndyn * (N + 1))
    qstc, qdyn = take(N), take(N)
    qvel, rvw, rww, qN, qthN, qrpd, accp, waccp = q[1], q[3], q[4], q[5], q[6], q[7], q[8], q[9]

    def rk4(s, a):                       # unicycle, one RK4 step (motion_model.py:141-163)
        f = lambda st: cs.vertcat(ts * a[0] * cs.cos(st[2]), ts * a[0] * cs.sin(st[2]), ts * a[1])
        k1 = f(s); k2 = f(s + 0.5 * k1); k3 = f(s + 0.5 * k2); k4 = f(s + k3)
        return s + (k1 + 2 * k2 + 2 * k3 + k4) / 6

    def seg_d2(px, py, ax, ay, bx, by):
        dx, dy = bx - ax, by - ay
        t = cs.fmin(cs.fmax(((px - ax) * dx + (py - ay) * dy) / (dx * dx + dy * dy + 1e-16), 0), 1)
        return (ax + t * dx - px) ** 2 + (ay + t * dy - py) ** 2

    def ell(px, py, e, rx, ry):          # 1 - rotated normalised distance (mpc_helper.py:38-52)
        dx, dy = px - e[0], py - e[1]
        ca, sa = cs.cos(e[4]), cs.sin(e[4])
        return 1 - ((dx * ca + dy * sa) / (rx + 1e-6)) ** 2 - ((dx * sa - dy * ca) / (ry + 1e-6)) ** 2

    cost, pen_s = 0, 0
    pen = [0] * Ndyn
    s = s0
    ref = [rs[3 * i:3 * i + 2] for i in range(N)] + [rs[3 * (N - 1):3 * (N - 1) + 2]]
    for k in range(N):
        a = u[2 * k:2 * k + 2]
        s = rk4(s, a)
        x, y = s[0], s[1]
        d2 = [seg_d2(x, y, ref[i][0], ref[i][1], ref[i + 1][0], ref[i + 1][1]) for i in range(k, N)]
        cost += qrpd * cs.mmin(cs.vertcat(*d2))
        cost += qvel * (a[0] - rv[k]) ** 2 + rvw * a[0] ** 2 + rww * a[1] ** 2
        for j in range(1, Nother):       # t = 0 fleet set skips robot 0 (mpc_builder.py:86-90)
            cost += 1000 * cs.fmax(0, width ** 2 - (x - c0[3 * j]) ** 2 - (y - c0[3 * j + 1]) ** 2)
        for j in range(Nother):
            b = 3 * (j * N + k)
            cost += 10 * cs.fmax(0, width ** 2 - (x - c[b]) ** 2 - (y - c[b + 1]) ** 2)
        for i in range(Nstc):
            o12 = os_[12 * i:12 * i + 12]
            ind = 1
            for e in range(4):
                ind *= cs.fmax(0, o12[e] - o12[4 + e] * x - o12[8 + e] * y)
            cost += qstc[k] * ind ** 2
            pen_s += cs.fmax(0, ind)
        for j in range(Ndyn):
            e0 = od[6 * ((N + 1) * j):6 * ((N + 1) * j) + 6]
            ek = od[6 * ((N + 1) * j + k + 1):6 * ((N + 1) * j + k + 1) + 6]
            pen[j] += cs.fmax(0, ell(x, y, e0, e0[2], e0[3])) + cs.fmax(0, ell(x, y, ek, ek[2], ek[3]))
            cost += 1000 * e0[5] * cs.fmax(0, ell(x, y, e0, e0[2] + vm + sm, e0[3] + vm + sm)) ** 2
            cost += qdyn[k] * ek[5] * cs.fmax(0, ell(x, y, ek, ek[2] + vm, ek[3] + vm)) ** 2
    cost += qN * ((s[0] - sN[0]) ** 2 + (s[1] - sN[1]) ** 2) + qthN * (s[2] - sN[2]) ** 2
    v, w = u[0::2], u[1::2]
    acc = (v - cs.vertcat(um1[0], v[:-1])) / ts
    wacc = (w - cs.vertcat(um1[1], w[:-1])) / ts
    cost += accp * cs.sumsqr(acc) + waccp * cs.sumsqr(wacc)
    f1 = cs.vertcat(acc, wacc)
    f2 = cs.vertcat(*[pen_s + pj for pj in pen])
    U = og.constraints.Rectangle([vmin, -wmax] * N, [vmax, wmax] * N)
    C = og.constraints.Rectangle([amin] * N + [-wamax] * N, [amax] * N + [wamax] * N)
    problem = og.builder.Problem(u, p, cost).with_constraints(U).with_aug_lagrangian_constraints(f1, C) \
        .with_penalty_constraints(f2)
    return problem, npar


def build_solver(build_dir, name="nmpc_open_check", max_duration_us=100_000, **dims):
    import opengen as og
    problem, npar = build_problem(**dims)
    cfg = og.config.BuildConfiguration().with_build_directory(build_dir).with_build_mode("release") \
        .with_build_python_bindings()
    meta = og.config.OptimizerMeta().with_optimizer_name(name)
    scfg = og.config.SolverConfiguration().with_initial_penalty(10).with_max_duration_micros(max_duration_us)
    og.builder.OpEnOptimizerBuilder(problem, meta, cfg, scfg).build()
    sys.path.insert(0, os.path.join(build_dir, name))
    return __import__(name).solver(), npar


def time_genuine_open(layout, P_host, cores, probe):
    """cpu_baseline record of kind "opengen": build the solver, run a bounded sample one problem at a time (the
    reference's own calling convention, trajectory_tracker.py:362), and compare with the oracle on the same instances --
    which is what pins (or refutes) the oracle's PANOC / ALM restatement."""
    import oracle
    build_dir = tempfile.mkdtemp(prefix="nmpc_open_")
    solver, npar = build_solver(build_dir, N=layout.N, Nother=layout.Nother, Nstc=layout.Nstc, Ndyn=layout.Ndyn)
    assert npar == layout.np_
    n = min(P_host.shape[0], 128)
    t0 = time.perf_counter()
    sols = [solver.run([float(v) for v in P_host[i]]) for i in range(n)]
    wall = time.perf_counter() - t0
    U = np.array([s.solution for s in sols])
    status = np.array([0 if s.exit_status == "Converged" else 1 for s in sols])
    pr = oracle.Problem(layout.N, layout.Nother, layout.Nstc, layout.Ndyn)
    rows = {}
    for form in (0, 1):
        Uo, ro = oracle.solve_batch(pr, oracle.Options(akkt_form=form), P_host[:n], nthreads=cores)
        both = (ro["status"] == 0) & (status == 0)
        du = np.abs(U - Uo).max(axis=1)
        rows[f"oracle_akkt_form_{form}"] = {
            "same_status_frac": float(np.mean(ro["status"] == status)), "both_converged": int(both.sum()),
            "median_abs_du_both_converged": float(np.median(du[both])) if both.any() else None,
            "inner_iters_oracle_mean": float(ro["inner_iters"].mean()),
            "inner_iters_open_mean": float(np.mean([s.num_inner_iterations for s in sols]))}
    return {"value": n / wall, "unit": "solves/s", "cores": 1, "kind": "opengen",
            "sample": f"first {n} instances of the timed batch, genuine OpEn solver (solver.run per instance, {wall:.2f} s wall)",
            "solve_time_ms_mean": float(np.mean([s.solve_time_ms for s in sols])), "oracle_vs_open": rows,
            "open_probe": "available", "open_probe_detail": probe}


if __name__ == "__main__":
    d = tempfile.mkdtemp(prefix="nmpc_open_")
    s, npar = build_solver(d)
    print("built", d, "np =", npar)
