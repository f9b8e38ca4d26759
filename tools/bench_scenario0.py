#!/usr/bin/env python3
"""BASELINE configs[0] in the reference's own timing protocol: the scenario_0 closed loop (1 robot, the scenario's
pedestrian + a second one, constant-velocity predictions, the warehouse's static map as the reference's map pipeline
extracts it, mpc_default.yaml) through the drop-in classes (MpcInterface -> TrajectoryTracker -> solver().run(p)), wall
time around every run_step, first 10 samples dropped, mean / max reported (main_base.py:309-318, 487-488).
One JSON line.   usage: bench_scenario0.py [fp32|fp64] [polish]"""
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import dyobav_mpcnwta_warehouse_amd as nm
from dyobav_mpcnwta_warehouse_amd import evaluate as ev
from dyobav_mpcnwta_warehouse_amd.mpc_interface import MpcInterface


def penetration(p, quads):
    """depth of p inside the closest-fitting quadrilateral (0 outside all): distance to the nearest edge"""
    a = quads
    b = np.roll(quads, -1, axis=1)
    ab = b - a
    cross = ab[..., 0] * (p[1] - a[..., 1]) - ab[..., 1] * (p[0] - a[..., 0])
    inside = (cross > 0).all(axis=1) | (cross < 0).all(axis=1)
    if not inside.any():
        return 0.0
    d = np.abs(cross) / np.hypot(ab[..., 0], ab[..., 1])
    return float(d[inside].min(axis=1).max())


def run(max_steps=160, with_map=True, second_pedestrian=False, swap_solver=None):
    """``swap_solver``: callable(tracker) -> object with ``run(parameters)`` that takes the place of the tracker's solver
    (tests/test_gpu_evaluate_reference.py puts the CPU oracle there to compare closed loops); the returned record then also
    carries the robot's trajectory."""
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "evaluate_cases.json")))
    sc = cases["scenario_0"]
    node = lambda k: tuple(sc["nodes_world"][str(k)])
    robot_path = [node(k) for k in sc["robot_path"]]
    start = np.array(sc["robot_start_world"])
    polys = cases["scenario_0_map"]["polygons_world"] if with_map else \
        [[(60.0 + i, 60.0), (59.5 + i, 60.0), (59.5 + i, 59.5), (60.0 + i, 59.5)] for i in range(12)]
    quads = np.array(polys, dtype=float)
    mi = MpcInterface("mpc_default.yaml", start.copy(), types.SimpleNamespace(processed_obstacle_list=[list(map(tuple, q)) for q in polys]),
                      verbose=False)
    mi.update_global_path(robot_path)
    if swap_solver is not None:
        mi.traj_tracker.solver = swap_solver(mi.traj_tracker)
    h0 = np.array(sc["human_starts_world"][0])
    # the scenario's own pedestrian (main_base.py:38-44); a second one walking the other way is optional (far away if off)
    hstart = np.array([[h0, np.array(node(32)) + np.array([-2.0, 0.0]) if second_pedestrian else np.array([40.0, 40.0])]])
    p0 = [node(k) for k in sc["human_paths"][0]]
    p1 = [node(32), node(9), node(9)] if second_pedestrian else [(40.0, 41.0), (40.0, 42.0), (40.0, 43.0)]
    e = ev.BatchEvaluator(nm.default_config_struct(), np.zeros((1, 3)), [[(5.0, 0.0)]], hstart, np.array([[p0, p1]]),
                          np.array([[[50.0, 50.0], [49.0, 50.0], [49.0, 49.0], [50.0, 49.0]]]), dtype=np.float64)
    state = start.copy()
    goal = np.array(robot_path[-1])
    times, min_ped, pen, traj = [], np.inf, 0.0, [state.copy()]
    for step in range(max_steps):
        rows = e._predict_cv().cpu().numpy()[0]
        mi.set_current_state(state)
        t0 = time.perf_counter()
        actions, pred, cost, closest, refs = mi.run_step("work", rows.tolist(), True)
        times.append(time.perf_counter() - t0)
        state = mi.state.copy()
        traj.append(state.copy())
        e._step_humans()
        peds = e.humans.cpu().numpy()[0]
        min_ped = min(min_ped, float(np.hypot(*(peds - state[:2]).T).min()))
        pen = max(pen, penetration(state[:2], quads))
        if np.abs(state[:2] - goal).max() <= 0.5:
            break
    e.close()
    t = np.array(times[10:]) * 1e3
    return {"metric": "closed-loop run_step wall time, scenario_0 (BASELINE configs[0]), reference protocol: first 10 samples dropped",
            "steps": len(times), "reached_goal": bool(np.abs(state[:2] - goal).max() <= 0.5), "mean_ms": float(t.mean()),
            "max_ms": float(t.max()), "median_ms": float(np.median(t)), "static_map_polygons": int(len(polys)) if with_map else 0,
            "pedestrians": 2 if second_pedestrian else 1,
            "max_penetration_into_an_inflated_polygon_m": pen, "min_pedestrian_distance_m": min_ped, "final_state": state.tolist(),
            **({"trajectory": np.array(traj).tolist()} if swap_solver is not None or os.environ.get("SCENARIO0_TRAJ") else {})}


if __name__ == "__main__":
    print(json.dumps(run()))
