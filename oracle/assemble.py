"""CPU ORACLE (test infrastructure) for the "next" row f1: assembling the solver's parameter vector from structured
inputs. numpy restatement of

* ``lineseg_dists``                        /root/reference/src/pkg_mpc_tracker/utils_geo.py:6-33
* ``polygon_halfspace_representation``     utils_geo.py:35-62   (for convex quadrilaterals given in vertex order; the
                                           reference runs scipy's qhull, whose facet order is arbitrary)
* ``MpcInterface.get_closest_n_stc_obstacles / get_stc_constraints / get_dyn_constraints``
                                           /root/reference/src/interfaces/mpc_interface.py:73-100
* the parameter concatenation of ``TrajectoryTracker.run_step``   pkg_mpc_tracker/trajectory_tracker.py:291-317

Pinned by ``tests/golden/assemble_cases.json`` (recorded by driving the reference's MpcInterface.run_step).
"""
from __future__ import annotations

import numpy as np


def lineseg_dists(p, a, b):
    """Distance from point p[2] to each segment a[i] -> b[i]."""
    d_ba = b - a
    d = d_ba / np.hypot(d_ba[:, 0], d_ba[:, 1])[:, None]
    s = ((a - p) * d).sum(axis=1)
    t = ((p - b) * d).sum(axis=1)
    h = np.maximum(np.maximum(s, t), 0.0)
    d_pa = p - a
    c = d_pa[:, 0] * d[:, 1] - d_pa[:, 1] * d[:, 0]
    return np.hypot(h, c)


def polygon_distance(p, poly):
    poly = np.asarray(poly, dtype=float)
    return float(np.min(lineseg_dists(np.asarray(p, dtype=float), poly, np.roll(poly, -1, axis=0))))


def quad_halfspaces(poly):
    """(b[4], a0[4], a1[4]) with row e built from the edge (v_e, v_{e+1}): A (x - centre) <= 1 inside."""
    poly = np.asarray(poly, dtype=float)
    centre = poly.mean(axis=0)
    V = poly - centre
    b, a0, a1 = [], [], []
    for e in range(len(poly)):
        F = np.stack([V[e], V[(e + 1) % len(poly)]])
        if np.linalg.matrix_rank(F) < 2:
            continue
        a = np.linalg.solve(F, np.ones(2))
        a0.append(a[0])
        a1.append(a[1])
        b.append(a @ centre + 1.0)
    return np.array(b), np.array(a0), np.array(a1)


def closest_polygons(state_xy, polys, n):
    """Indices of the n polygons closest to the robot, nearest first (the reference's argpartition returns the same
    set in unspecified order)."""
    d = np.array([polygon_distance(state_xy, q) for q in polys])
    return np.argsort(d, kind="stable")[:n]


def assemble(last_u, state, ref_states, speed_ref, tuning, other_robots, map_polygons, dyn_obstacles, stc_weights,
             dyn_weights, N=20, Nother=10, Nstc=10, Ndyn=15):
    """One parameter vector p (SURVEY.md 8a layout)."""
    os_ = np.zeros(12 * Nstc)
    for slot, m in enumerate(closest_polygons(state[:2], map_polygons, Nstc)):
        b, a0, a1 = quad_halfspaces(map_polygons[m])
        os_[12 * slot:12 * slot + 12] = np.concatenate([b, a0, a1])
    od = np.zeros(Ndyn * (N + 1) * 6)
    if dyn_obstacles is not None and len(dyn_obstacles):
        flat = np.asarray(dyn_obstacles, dtype=float).reshape(-1)
        od[:flat.size] = flat
    other = np.zeros(3 * (N + 1) * Nother) if other_robots is None else np.asarray(other_robots, dtype=float)
    ref_states = np.asarray(ref_states, dtype=float)
    return np.concatenate([np.asarray(last_u, float), np.asarray(state, float), ref_states[-1], np.asarray(tuning, float),
                           ref_states.reshape(-1), np.full(N, float(speed_ref)), other, os_, od,
                           np.asarray(stc_weights, float), np.asarray(dyn_weights, float)])


def canonical_static_block(os_block, Nstc=10):
    """Order-free view of the o_s block: per polygon the (b, a0, a1) rows sorted, polygons sorted -- for comparing
    against the reference, whose polygon order (argpartition) and facet order (qhull) are unspecified."""
    polys = []
    for s in range(Nstc):
        q = np.asarray(os_block[12 * s:12 * s + 12], dtype=float)
        rows = sorted(zip(q[0:4].round(9), q[4:8].round(9), q[8:12].round(9)))
        polys.append(tuple(rows))
    return sorted(polys)
