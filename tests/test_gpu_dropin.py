"""GPU tests of the drop-in surface: ``solver().run(p)`` and ``TrajectoryTracker.run_step`` on the HIP solver,
checked against the same harness driven by the CPU oracle."""
import os
import sys
import types

import numpy as np
import pytest

import dyobav_mpcnwta_warehouse_amd as nm
import oracle
from dyobav_mpcnwta_warehouse_amd import solver_build
from dyobav_mpcnwta_warehouse_amd.configs import CircularRobotSpecification, MpcConfiguration
from dyobav_mpcnwta_warehouse_amd.motion_model import UnicycleModel
from dyobav_mpcnwta_warehouse_amd.solver import Solver, make_config
from dyobav_mpcnwta_warehouse_amd.trajectory_tracker import TrajectoryTracker

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "config", "mpc_fast.yaml")


class _OracleSolver:
    """Test-only adapter: the oracle behind the same run() surface (stateful multipliers like Solver)."""

    def __init__(self, **opts):
        self.pr, self.op = oracle.Problem(), oracle.Options(**opts)
        self.y = np.zeros(40)

    def run(self, p, initial_guess=None, initial_lagrange_multipliers=None, initial_penalty=None):
        u, y, res = oracle.solve(self.pr, self.op, np.asarray(p, dtype=np.float64), u0=initial_guess, y0=self.y)
        self.y = y
        return types.SimpleNamespace(solution=u.tolist(), cost=float(res["cost"]),
                                     exit_status=oracle.STATUS_NAMES[int(res["status"])], solve_time_ms=0.0)


def test_solver_run_surface_and_errors():
    mpc, rob = MpcConfiguration.from_yaml(CFG), CircularRobotSpecification.from_yaml(CFG)
    s = Solver(make_config(mpc, rob))
    assert s.num_parameters == 2778 and s.num_decision_variables == 40
    p = nm.scenarios.make_batch(1, nm.scenarios.ParamLayout(), seed=4, n_ped=0, n_boxes=0)[0]
    sol = s.run(p.tolist())                       # flat Python list, as the tracker passes it
    assert sol.exit_status in nm.EXIT_STATUS_NAMES[:2] and len(sol.solution) == 40
    assert sol.num_outer_iterations >= 2 and sol.solve_time_ms > 0 and len(sol.lagrange_multipliers) == 40
    f, _, _ = oracle.eval_problem(oracle.Problem(), np.array(sol.solution), p)
    assert sol.cost == pytest.approx(f, rel=1e-9)                       # cost = f(u*) without penalty terms
    assert s.run(p[:-1].tolist()) is None                              # OpEn: wrong length -> None
    assert s.run(p.tolist(), initial_guess=[0.0] * 39) is None
    s.close()


def test_closed_loop_tracker_matches_oracle_backed_tracker():
    mpc, rob = MpcConfiguration.from_yaml(CFG), CircularRobotSpecification.from_yaml(CFG)
    opts = dict(lip_delta=1e-4, lip_eps=1e-4, max_outer=3, max_inner=30)
    gpu_cfg = make_config(mpc, rob, lip_delta_f64=1e-4, lip_eps_f64=1e-4, max_outer_iterations=3,
                          max_inner_iterations=30)
    trackers = [TrajectoryTracker(mpc, rob, solver_factory=lambda: Solver(gpu_cfg)),
                TrajectoryTracker(mpc, rob, solver_factory=lambda: _OracleSolver(**opts))]
    ped = [[2.5 - 0.2 * t, 0.6, 0.2 + 0.05 * t, 0.2 + 0.05 * t, 0, 1] for t in range(21)]
    dyn = [v for row in ped for v in row] + [0.0] * (14 * 21 * 6)
    logs = []
    for tr in trackers:
        tr.load_motion_model(UnicycleModel(rob.ts))
        tr.load_init_states(np.array([0.0, 0.0, 0.0]), np.array([6.0, 0.0, 0.0]))
        tr.set_work_mode("work")
        tr.set_ref_trajectory([(6.0, 0.0)])
        acts = []
        for _ in range(4):
            actions, pred_states, ref_states, cost = tr.run_step(None, dyn, mode="work")
            acts.append(actions[0])
        logs.append(np.array(acts))
    assert np.abs(logs[0] - logs[1]).max() < 1e-6
    assert (logs[0][:, 0] > 0).all()


def test_generated_module_is_importable_by_the_reference_convention(tmp_path):
    """What reference trajectory_tracker.py:58-61 does: sys.path.append(<build_dir>/<name>); __import__(name).solver()"""
    mod = solver_build.build(CFG, out_dir=str(tmp_path), compile_library=False)
    sys.path.append(os.path.dirname(mod))
    try:
        s = __import__("navi_fast").solver()
        p = nm.scenarios.make_batch(1, nm.scenarios.ParamLayout(), seed=5, n_ped=0, n_boxes=0)[0]
        sol = s.run(p.tolist())
        assert sol is not None and len(sol.solution) == 40 and isinstance(sol.exit_status, str)
    finally:
        sys.path.remove(os.path.dirname(mod))
        sys.modules.pop("navi_fast", None)


def test_mpc_interface_matches_reference_recording(golden_dir):
    """L3 adapter: MpcInterface.run_step with the static-obstacle marshalling on the device, against the recording of
    the reference's MpcInterface.run_step (parameter vector handed to the solver + closest_obstacle_list)."""
    import json
    from dyobav_mpcnwta_warehouse_amd.mpc_interface import MpcInterface
    from oracle import assemble as oa

    class Fake:
        def __init__(self):
            self.calls = []

        def run(self, p, *a, **k):
            self.calls.append([float(v) for v in p])
            return types.SimpleNamespace(solution=[0.3, 0.05] * 20, cost=1.0, exit_status="Converged", solve_time_ms=1.0)

    for c in json.load(open(os.path.join(golden_dir, "assemble_cases.json"))):
        fake = Fake()
        geo = types.SimpleNamespace(processed_obstacle_list=[[tuple(v) for v in q] for q in c["map_polygons"]])
        mi = MpcInterface("mpc_fast.yaml", np.array(c["state"]), geo, verbose=False, solver_factory=lambda: fake)
        mi.update_global_path([tuple(c["goal"])])
        actions, pred, cost, closest, refs = mi.run_step("work", c["dyn"] if len(c["dyn"]) else None, True)
        p, p_ref = np.array(fake.calls[-1]), np.array(c["params"])
        np.testing.assert_allclose(p[:728], p_ref[:728], rtol=0, atol=1e-12)
        np.testing.assert_allclose(p[848:], p_ref[848:], rtol=0, atol=1e-12)
        np.testing.assert_allclose(np.array(oa.canonical_static_block(p[728:848])),
                                   np.array(oa.canonical_static_block(p_ref[728:848])), rtol=0, atol=1e-7)
        key = lambda polys: sorted(tuple(map(tuple, np.round(np.array(q), 9))) for q in polys)
        assert key(closest) == key(c["closest"])
        np.testing.assert_allclose(np.asarray(refs), np.array(c["ref_states"]), atol=1e-12)
        assert len(actions) == 1 and len(pred) == 20 and cost == 1.0


def test_config0_like_closed_loop_one_robot_two_pedestrians():
    """BASELINE configs[0] (functional case): warehouse-like loop, 1 robot, 2 pedestrians on constant-velocity
    predictions, static boxes, mpc_default.yaml, GPU solver behind MpcInterface/TrajectoryTracker. The robot makes
    progress along the path, never enters a static box or a pedestrian disc, and every solve returns a legal status."""
    from dyobav_mpcnwta_warehouse_amd.mpc_interface import MpcInterface
    boxes = [[(3.0, 1.2), (2.0, 1.2), (2.0, 0.6), (3.0, 0.6)], [(5.5, -0.6), (4.5, -0.6), (4.5, -1.4), (5.5, -1.4)],
             [(8.0, 1.5), (7.0, 1.5), (7.0, 0.7), (8.0, 0.7)]] + \
            [[(20.0 + i, 20.0), (19.5 + i, 20.0), (19.5 + i, 19.5), (20.0 + i, 19.5)] for i in range(9)]
    geo = types.SimpleNamespace(processed_obstacle_list=boxes)
    mi = MpcInterface("mpc_default.yaml", np.array([0.0, 0.0, 0.0]), geo, verbose=False)
    mi.update_global_path([(10.0, 0.0)])
    peds = np.array([[6.0, 2.5], [9.0, -2.0]])
    pvel = np.array([[-0.6, -0.5], [-0.8, 0.35]])
    state = np.array([0.0, 0.0, 0.0])
    traj, stats = [state.copy()], []
    for step in range(40):
        dyn = [[[float(p[0] + v[0] * 0.2 * t), float(p[1] + v[1] * 0.2 * t), 0.2 + 0.03 * t, 0.2 + 0.03 * t, 0, 1]
                for t in range(21)] for p, v in zip(peds, pvel)]
        mi.set_current_state(state)
        actions, pred, cost, closest, refs = mi.run_step("work", dyn, True)
        assert len(closest) == 10 and np.isfinite(cost)
        state = mi.state.copy()
        peds = peds + pvel * 0.2
        traj.append(state.copy())
        for p in peds:
            assert np.hypot(*(state[:2] - p)) > 0.2          # main_pre.check_collision: HUMAN_SIZE
        for q in boxes[:3]:
            xs, ys = [v[0] for v in q], [v[1] for v in q]
            assert not (min(xs) < state[0] < max(xs) and min(ys) < state[1] < max(ys))
    traj = np.array(traj)
    assert traj[-1, 0] > 5.0 and np.abs(traj[:, 1]).max() < 2.5
    assert len(mi.traj_tracker.past_actions) == 40
