"""CPU tests of the host-side mirror of the reference interface (configs, motion model, TrajectoryTracker harness,
solver_build), pinned by fixtures recorded from the reference's own classes (tests/golden/make_golden.py)."""
import json
import os
import sys
import types

import numpy as np
import pytest

from dyobav_mpcnwta_warehouse_amd import solver_build
from dyobav_mpcnwta_warehouse_amd.configs import CircularRobotSpecification, MpcConfiguration
from dyobav_mpcnwta_warehouse_amd.motion_model import UnicycleModel, unicycle_model
from dyobav_mpcnwta_warehouse_amd.trajectory_tracker import TrajectoryTracker

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
CFG = os.path.join(ROOT, "config", "mpc_fast.yaml")


def test_yaml_surface_matches_reference_keys():
    """Same keys / values as the reference's config files (values recorded in problem_meta.json + SURVEY 8a)."""
    for name, opt, tmax in (("mpc_fast.yaml", "navi_fast", 100_000), ("mpc_default.yaml", "navi_default", 500_000)):
        path = os.path.join(ROOT, "config", name)
        mpc, rob = MpcConfiguration.from_yaml(path), CircularRobotSpecification.from_yaml(path)
        assert (mpc.N_hor, mpc.ns, mpc.nu, mpc.nq, mpc.Nother, mpc.Nstcobs, mpc.nstcobs, mpc.Ndynobs, mpc.ndynobs) == \
            (20, 3, 2, 10, 10, 10, 12, 15, 6)
        assert mpc.optimizer_name == opt and mpc.max_solver_time == tmax and mpc.build_directory == "mpc_solver"
        assert mpc.bad_exit_codes == ["NotConvergedIterations", "NotConvergedOutOfTime"]
        assert (rob.lin_vel_min, rob.lin_vel_max, rob.ang_vel_max) == (-0.5, 1.5, 0.5)
        assert (rob.lin_acc_min, rob.lin_acc_max, rob.ang_acc_max) == (-1, 1, 3)
        assert (rob.vehicle_width, rob.vehicle_margin, rob.social_margin, rob.ts) == (0.5, 0.2, 0.2, 0.2)
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "problem_meta.json")))
    assert meta["umin"][:2] == [-0.5, -0.5] and meta["umax"][:2] == [1.5, 0.5]
    assert meta["cmin"][0] == -1 and meta["cmax"][-1] == 3 and meta["np"] == 2778


def test_motion_model_matches_reference(golden_dir):
    fx = np.load(os.path.join(golden_dir, "motion_model.npz"))
    m = UnicycleModel(float(fx["ts"]))
    for s, a, sn in zip(fx["S"], fx["A"], fx["S_next"]):
        np.testing.assert_allclose(m(s, a), sn, rtol=0, atol=1e-14)
        np.testing.assert_allclose(unicycle_model(s, a, float(fx["ts"])), sn, rtol=0, atol=1e-14)


class _ScriptedSolver:
    """Same scripted replies as the fake solver the fixture was recorded with (make_golden.tracker_harness)."""

    def __init__(self):
        self.calls = []

    def run(self, p, initial_guess=None, initial_lagrange_multipliers=None, initial_penalty=None):
        self.calls.append(dict(p=[float(v) for v in p], kw=(initial_guess, initial_lagrange_multipliers,
                                                             initial_penalty)))
        n = len(self.calls)
        u = []
        for k in range(20):
            u += [0.5 + 0.02 * k + 0.1 * n, 0.1 - 0.01 * k]
        return types.SimpleNamespace(solution=u, cost=12.5 * n, exit_status="Converged", solve_time_ms=1.25)


def test_tracker_harness_matches_reference_recording(golden_dir):
    rec = json.load(open(os.path.join(golden_dir, "tracker_harness.json")))
    mpc, rob = MpcConfiguration.from_yaml(CFG), CircularRobotSpecification.from_yaml(CFG)
    fake = _ScriptedSolver()
    tr = TrajectoryTracker(mpc, rob, verbose=False, solver_factory=lambda: fake)
    tr.load_motion_model(UnicycleModel(rob.ts))
    tr.load_init_states(np.array([0.0, 0.0, 0.0]), np.array([5.0, 3.0, 1.57]))
    tr.set_work_mode("work")
    tr.set_ref_trajectory([(5.0, 0.0), (5.0, 3.0)])
    np.testing.assert_allclose(np.array(tr.ref_traj), np.array(rec["ref_traj"]), rtol=0, atol=1e-12)
    for i, st in enumerate(rec["steps"]):
        if i == 3:
            tr.set_current_state(np.array([4.9, 2.6, 1.5]))
        np.testing.assert_allclose(tr.state, st["state_in"], atol=1e-12)
        actions, pred_states, ref_states, cost = tr.run_step(st["stc"], st["dyn"], mode="work")
        call = fake.calls[-1]
        assert len(call["p"]) == 2778 and call["kw"] == (None, None, None)
        np.testing.assert_allclose(call["p"], st["params"], rtol=0, atol=1e-12)   # order and values of all 12 blocks
        np.testing.assert_allclose(np.array(actions), np.array(st["actions"]), atol=1e-12)
        np.testing.assert_allclose(np.array(pred_states), np.array(st["pred_states"]), atol=1e-12)
        np.testing.assert_allclose(ref_states, np.array(st["ref_states"]), atol=1e-12)
        assert cost == st["cost"] and tr.idx_ref_traj == st["idx_ref_traj"]
        np.testing.assert_allclose(tr.state, st["state_out"], atol=1e-12)
        assert tr.base_speed == pytest.approx(st["base_speed"])
    # quirk: near the goal the speed reference jumps to lin_vel_max (max(), reference :308-309)
    L_rv = 18 + 60
    assert rec["steps"][3]["params"][L_rv] == 1.5 and rec["steps"][0]["params"][L_rv] == pytest.approx(1.2)
    np.testing.assert_allclose(np.array(tr.past_actions), np.array(rec["past_actions"]), atol=1e-12)
    np.testing.assert_allclose(np.array(tr.past_states), np.array(rec["past_states"]), atol=1e-12)
    assert tr.cost_timelist == rec["cost_timelist"] and tr.solver_time_timelist == rec["solver_time_timelist"]
    case = rec["get_ref_traj_case"]
    out = TrajectoryTracker.get_ref_traj(case["ts"], [tuple(p) for p in case["path"]], tuple(case["state"]), case["speed"])
    np.testing.assert_allclose(np.array(out), np.array(case["out"]), atol=1e-12)


def test_tracker_argument_errors_match_reference():
    mpc, rob = MpcConfiguration.from_yaml(CFG), CircularRobotSpecification.from_yaml(CFG)
    tr = TrajectoryTracker(mpc, rob, solver_factory=_ScriptedSolver)
    with pytest.raises(TypeError):
        tr.load_init_states([0, 0, 0], np.zeros(3))
    with pytest.raises(TypeError):
        tr.set_current_state((0, 0, 0))
    with pytest.raises(ModuleNotFoundError):
        tr.set_work_mode("warp")
    with pytest.raises(TypeError):
        tr.set_obstacle_weights("10", 10)
    tr.set_work_mode("aligning")
    assert tr.base_speed == 0.75 and tr.tuning_params == [0.0, 0.0, 100, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]


def test_solver_build_writes_importable_module(tmp_path):
    mod = solver_build.build(CFG, out_dir=str(tmp_path), compile_library=False)
    assert mod.endswith(os.path.join("mpc_solver", "navi_fast", "navi_fast.py"))
    sys.path.insert(0, os.path.dirname(mod))
    try:
        m = __import__("navi_fast")
        assert callable(m.solver) and m.YAML_PATH == CFG
    finally:
        sys.path.pop(0)
        sys.modules.pop("navi_fast", None)


def test_closed_loop_scenarios_are_deterministic_and_well_formed():
    """scenarios.make_closed_loop_scenarios (the inputs of the batched closed-loop evaluator, row f3, and of
    harvest_closed_loop): deterministic in (B, seed, n_ped); shapes; pedestrians start on either side of the aisle and
    cross it; the robot's goal lies ~8 m ahead. (No GPU: the generator is host-side numpy.)"""
    import numpy as np
    from dyobav_mpcnwta_warehouse_amd import scenarios
    a = scenarios.make_closed_loop_scenarios(50, seed=13, n_ped=4)
    b = scenarios.make_closed_loop_scenarios(50, seed=13, n_ped=4)
    c = scenarios.make_closed_loop_scenarios(50, seed=14, n_ped=4)
    for k in ("robot_starts", "human_starts", "human_paths", "map_polygons"):
        assert np.array_equal(a[k], b[k]), k
    assert a["robot_paths"] == b["robot_paths"] and not np.array_equal(a["human_starts"], c["human_starts"])
    assert a["robot_starts"].shape == (50, 3) and a["human_starts"].shape == (50, 4, 2)
    assert a["human_paths"].shape == (50, 4, 2, 2) and a["map_polygons"].shape == (14, 4, 2)
    assert len(a["robot_paths"]) == 50 and all(len(p) == 1 and 7.4 <= p[0][0] <= 8.6 for p in a["robot_paths"])
    hs, hp = a["human_starts"], a["human_paths"]
    assert np.all(np.abs(hs[..., 1]) >= 2.0) and np.all(np.sign(hp[:, :, 0, 1]) == -np.sign(hs[..., 1]))   # they cross the aisle
    assert np.all(hp[:, :, 0, 0] < hs[..., 0])                                                              # ... towards the robot
    # the shelf blocks line the aisle on both sides and leave it free
    cy = a["map_polygons"][:, :, 1].mean(axis=1)
    assert (cy > 0).sum() == 7 and (cy < 0).sum() == 7 and np.all(np.abs(a["map_polygons"][:, :, 1]).min(axis=1) >= 1.0)
    with __import__("pytest").raises(AssertionError):
        from dyobav_mpcnwta_warehouse_amd.scenarios import ParamLayout, make_batch
        make_batch(2, ParamLayout(20, 10, 10, 15), n_ped=4, n_hyp=10)       # 40 rows do not fit Ndynobs = 15


def test_reference_scenarios_are_the_reference_s_own_geometry():
    """scenarios.make_reference_scenarios (round 6): run b of the batch is the reference's scenario_{b % 3} -- start state, node
    path and pedestrian exactly as tests/golden/make_golden.py recorded them from main_base.py:36-58 through the reference's
    transform -- on the 55 static rectangles of its warehouse map; the other pedestrians walk seeded node paths of the same
    graph. Deterministic; a batch is a prefix of every larger batch with the same seed. The package's copy of the warehouse
    data is the golden file, byte for byte. (No GPU: the generator is host-side numpy.)"""
    import filecmp
    import numpy as np
    from dyobav_mpcnwta_warehouse_amd import scenarios
    pkg = os.path.join(ROOT, "dyobav-mpcnwta-warehouse_amd", "data", "warehouse_world.json")
    assert filecmp.cmp(pkg, os.path.join(GOLDEN, "warehouse_world.json"), shallow=False)
    ec = json.load(open(os.path.join(GOLDEN, "evaluate_cases.json")))
    w = scenarios.warehouse_world()
    assert w["scenarios"] == ec["scenarios"] and w["nodes_world"] == ec["scenario_0"]["nodes_world"]
    assert len(w["map_polygons_world"]) == 55 and len(w["nodes_world"]) == 32 and len(w["graph_edges"]) == 55
    assert w["constants"]["HUMAN_STAGGER"] == scenarios.HUMAN_STAGGER == 0.5 and w["constants"]["HUMAN_VMAX"] == 1.5
    a = scenarios.make_reference_scenarios(60, seed=13, n_ped=4)
    b = scenarios.make_reference_scenarios(24, seed=13, n_ped=4)
    c = scenarios.make_reference_scenarios(60, seed=14, n_ped=4)
    for k in ("robot_starts", "human_starts", "human_paths", "scenario_index"):
        assert np.array_equal(a[k][:24], b[k]), k
    assert a["robot_paths"][:24] == b["robot_paths"] and not np.array_equal(a["human_starts"], c["human_starts"])
    assert a["map_polygons"].shape == (55, 4, 2) and a["human_starts"].shape == (60, 4, 2) and a["human_paths"].shape == (60, 4, 4, 2)
    nodes = {tuple(v) for v in w["nodes_world"].values()}
    edges = {frozenset((tuple(w["nodes_world"][str(x)]), tuple(w["nodes_world"][str(y)]))) for x, y in w["graph_edges"]}
    for i in range(60):
        s = w["scenarios"][str(i % 3)]
        assert a["scenario_index"][i] == i % 3
        assert a["robot_starts"][i].tolist() == s["robot_start_world"]                    # main_base.py:136 ct2real(ROBOT_START_POINT)
        assert [list(p) for p in a["robot_paths"][i]] == s["robot_path_world"]             # :138 the node list, in world coordinates
        assert a["human_starts"][i, 0].tolist() == s["human_starts_world"][0]              # the scenario's own pedestrian ...
        own = s["human_paths_world"][0]
        assert a["human_paths"][i, 0].tolist() == (own + [own[-1]] * 4)[:4]                # ... on its own node path (padded)
        for h in range(1, 4):                                                              # the others: walks along graph edges
            path = [tuple(p) for p in a["human_paths"][i, h].tolist()]
            assert all(p in nodes for p in path)
            first = min(nodes, key=lambda n: np.hypot(n[0] - a["human_starts"][i, h, 0], n[1] - a["human_starts"][i, h, 1]))
            assert np.abs(np.array(first) - a["human_starts"][i, h]).max() <= 0.5
            walk = [first] + path
            assert all(frozenset((walk[j], walk[j + 1])) in edges for j in range(4))
            assert all(walk[j + 2] != walk[j] or len({frozenset((walk[j + 1], n)) for n in nodes} & edges) == 1 for j in range(3))
            assert np.hypot(*(a["human_starts"][i, h] - a["robot_starts"][i, :2])) >= 2.3   # not on top of the robot
    one = scenarios.make_reference_scenarios(9, seed=1, scenario=2)
    assert (one["scenario_index"] == 2).all() and one["robot_starts"][:, 0].tolist() == [w["scenarios"]["2"]["robot_start_world"][0]] * 9
