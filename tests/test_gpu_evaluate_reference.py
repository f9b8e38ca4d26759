"""f3 pinned to the reference: the pieces `evaluate.BatchEvaluator` restates, checked against recordings of the
reference's own classes (tests/golden/evaluate_cases.json, written by tests/golden/make_golden.py::evaluate_fixture):

  * pedestrian motion          basic_agent.Human.run_step               (basic_agent.py:52-82), stagger draws replayed
  * constant-velocity predictor interfaces/cvmp_interface.py:24-57
  * robot motion               basic_agent.Robot.one_step (unicycle RK4; motion_model.py:141-163)
  * evaluation metrics         main_pre.calc_action_smoothness / calc_minimal_dynamic_obstacle_distance /
                               calc_deviation_distance                  (main_pre.py:34-53)
  * BASELINE configs[0] on the real scenario: main_base.scenario_0 (main_base.py:38-44) with the node coordinates of
    data/warehouse_sim_original/mygraph.json in world coordinates, 1 robot, the scenario's pedestrian + a second one,
    mpc_default.yaml, GPU solver behind MpcInterface / TrajectoryTracker.
"""
import json
import os
import types

import numpy as np
import pytest
import torch

import dyobav_mpcnwta_warehouse_amd as nm
from dyobav_mpcnwta_warehouse_amd import evaluate as ev

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def cases():
    return json.load(open(os.path.join(GOLDEN, "evaluate_cases.json")))


def _evaluator(hstart, hpath, stagger=0.0):
    """A BatchEvaluator with B = len(hstart) scenarios whose robots stand still; only the pedestrian side is used."""
    B = hstart.shape[0]
    boxes = np.array([[[50.0, 50.0], [49.0, 50.0], [49.0, 49.0], [50.0, 49.0]]])
    starts = np.zeros((B, 3))
    paths = [[(5.0, 0.0)] for _ in range(B)]
    return ev.BatchEvaluator(nm.default_config_struct(), starts, paths, hstart, hpath, boxes, dtype=np.float64,
                             human_stagger=stagger)


def test_pedestrian_motion_matches_recorded_reference_walks(cases):
    walks = cases["human_walks"]
    hstart = np.array([[w["start"]] for w in walks])                          # [B,1,2]
    hpath = np.array([[w["path"]] for w in walks])                            # [B,1,3,2]
    e = _evaluator(hstart, hpath, stagger=0.5)
    T = len(walks[0]["moved"])
    e.stagger_replay = [torch.tensor([[w["stagger_draws"][t]] for w in walks], dtype=torch.float64, device=e.dev)
                        for t in range(T)]
    for t in range(T):
        e._step_humans()
        got = e.humans.cpu().numpy()[:, 0]
        want = np.array([w["states"][t + 1] for w in walks])
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-12, err_msg=f"step {t}")
    # the walks reach the end of their paths and stop, like the reference (run_step returns False)
    assert not all(w["moved"][-1] for w in walks)
    e.close()


def test_constant_velocity_prediction_matches_recorded_reference(cases):
    cv = cases["cv_cases"]
    B = len(cv)
    hstart = np.array([[c["traj"][-1]] for c in cv])
    e = _evaluator(hstart, np.zeros((B, 1, 1, 2)))
    hist = np.zeros((B, 1, 5, 2))
    count = np.zeros((B, 1), dtype=np.int64)
    for b, c in enumerate(cv):
        pts = np.array(c["traj"])[-5:]                                        # cvmp_interface.py:41: the latest 5 points
        hist[b, 0, 5 - len(pts):] = pts
        hist[b, 0, :5 - len(pts)] = pts[0]
        count[b, 0] = len(pts)
    e.hist = torch.as_tensor(hist, device=e.dev)
    e.hcount = torch.as_tensor(count, device=e.dev)
    rows = e._predict_cv().cpu().numpy()                                      # [B,1,N+1,6]
    for b, c in enumerate(cv):
        np.testing.assert_allclose(rows[b, 0, 0, :2], c["traj"][-1], atol=1e-12)
        np.testing.assert_allclose(rows[b, 0, 1:, :2], np.array(c["positions"]), rtol=0, atol=1e-12)
        assert (rows[b, 0, 1:, 2:4] == np.array(c["uncertainty"])).all() and (rows[b, 0, 0, 2:4] == ev.HUMAN_SIZE).all()
    e.close()


def test_robot_step_and_metrics_match_recorded_reference(cases):
    dev = torch.device("cuda")
    rs = cases["robot_steps"]
    nxt = ev.unicycle_rk4_step(torch.tensor([r["state"] for r in rs], dtype=torch.float64, device=dev),
                               torch.tensor([r["action"] for r in rs], dtype=torch.float64, device=dev), cases["ts"])
    np.testing.assert_allclose(nxt.cpu().numpy(), np.array([r["next"] for r in rs]), rtol=0, atol=1e-12)
    for m in cases["metric_cases"]:
        A = torch.tensor([m["actions"]], dtype=torch.float64, device=dev)
        np.testing.assert_allclose(ev.action_smoothness(A).cpu().numpy()[0], m["smoothness"], rtol=1e-12)
        st = torch.tensor([m["state"][:2]], dtype=torch.float64, device=dev)
        hum = torch.tensor([m["humans"]], dtype=torch.float64, device=dev)
        assert float(ev.min_dynamic_distance(st, hum)[0]) == pytest.approx(m["min_dyn_distance"], rel=1e-12)
        ref = torch.tensor([m["ref_traj"]], dtype=torch.float64, device=dev)
        act = torch.tensor(m["actual_traj"], dtype=torch.float64, device=dev)
        n = torch.tensor([ref.shape[1]], device=dev)
        d = torch.stack([ev.deviation_to_reference(p[None, :], ref, n)[0] for p in act])
        assert float(d.mean()) == pytest.approx(m["deviation"][0], rel=1e-12)
        assert float(d.max()) == pytest.approx(m["deviation"][1], rel=1e-12)


def test_config0_scenario_0_one_robot_two_pedestrians(cases):
    """BASELINE configs[0] on the reference's scenario_0 geometry (world coordinates recorded from the reference's
    transform): robot 16 -> 32 node path from its start, the scenario's pedestrian walking 9 -> 32 -> 16 (reference
    Human dynamics, replayed through the evaluator's pedestrian step with stagger 0) plus a second pedestrian coming the
    other way; constant-velocity predictions; mpc_default.yaml; GPU solver behind MpcInterface / TrajectoryTracker
    (the reference's own classes' mirror). The static map needs skimage / pyclipper / shapely (absent) and is left out
    (placeholder boxes far away). The robot follows the path, never touches a pedestrian and reaches the goal node."""
    from dyobav_mpcnwta_warehouse_amd.mpc_interface import MpcInterface
    sc = cases["scenario_0"]
    node = lambda k: tuple(sc["nodes_world"][str(k)])
    robot_path = [node(k) for k in sc["robot_path"]]
    start = np.array(sc["robot_start_world"])
    far = [[(60.0 + i, 60.0), (59.5 + i, 60.0), (59.5 + i, 59.5), (60.0 + i, 59.5)] for i in range(12)]
    mi = MpcInterface("mpc_default.yaml", start.copy(), types.SimpleNamespace(processed_obstacle_list=far), verbose=False)
    mi.update_global_path(robot_path)
    h0 = np.array(sc["human_starts_world"][0])
    hstart = np.array([[h0, np.array(node(32)) + np.array([-2.0, 0.0])]])                    # [1,2,2]
    p0 = [node(k) for k in sc["human_paths"][0]]
    p1 = [node(32), node(9), node(9)]
    e = _evaluator(hstart, np.array([[p0, p1]]))
    state = start.copy()
    min_dist, traj = np.inf, [state.copy()]
    goal = np.array(robot_path[-1])
    for step in range(120):
        rows = e._predict_cv().cpu().numpy()[0]                                              # [2, N+1, 6]
        mi.set_current_state(state)
        actions, pred, cost, closest, refs = mi.run_step("work", rows.tolist(), True)
        assert np.isfinite(cost) and len(actions) == 1
        state = mi.state.copy()
        e._step_humans()
        peds = e.humans.cpu().numpy()[0]
        min_dist = min(min_dist, float(np.hypot(*(peds - state[:2]).T).min()))
        traj.append(state.copy())
        if np.abs(state[:2] - goal).max() <= 0.5:                                            # main_base.py:338-340
            break
    e.close()
    traj = np.array(traj)
    assert min_dist > ev.HUMAN_SIZE, min_dist                                                # main_pre.check_collision
    assert np.abs(traj[-1, :2] - goal).max() <= 0.5, traj[-1]
    assert step < 119


def test_config0_scenario_0_with_the_warehouse_static_map(cases):
    """The same closed loop with the warehouse's real static map in it: the 55 inflated rectangles the reference's own
    map pipeline (MapInterface.cvt_occ2geo, main_base.py:123-127) extracts from data/warehouse_sim_original/mymap.pgm --
    recorded by tests/golden/make_golden.py with stand-ins for skimage.find_contours / pyclipper -- so that the
    closest-polygon selection, the half-space rows and the static-obstacle terms see BASELINE configs[0]'s obstacles.
    Also the reference's timing protocol for this loop (tools/bench_scenario0.py; first 10 samples dropped)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), "..", "tools"))
    import bench_scenario0
    assert len(cases["scenario_0_map"]["polygons_world"]) == 55
    r = bench_scenario0.run(max_steps=160, with_map=True)
    print(json.dumps(r))
    # static obstacles are soft terms of this MPC: dodging the pedestrian who comes down the same 1.7 m corridor the robot
    # may clip the inflated corner of a shelf; it must not get near the shelf itself (robot radius 0.25 m)
    assert r["reached_goal"] and r["max_penetration_into_an_inflated_polygon_m"] < 0.15, r
    assert r["min_pedestrian_distance_m"] > ev.HUMAN_SIZE, r
    assert r["steps"] < 160 and r["mean_ms"] < 100.0          # (the reference's budget per step: max_solver_time = 0.1 s)


def test_config0_scenario_0_driven_by_the_oracle_against_the_kernels(cases):
    """BASELINE configs[0] as a system-level parity statement: the scenario_0 closed loop with the warehouse's static map,
    through the drop-in interface (MpcInterface -> TrajectoryTracker.run_step -> solver.run(p), multipliers carried between
    calls as OpEn's binding does), driven once by the HIP kernels and once by the CPU oracle put in the tracker's solver
    slot -- and by the oracle's re-associated twin as the yardstick. Same options on both sides (what ``make_config``
    derives from mpc_default.yaml), no wall-clock budget."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), "..", "tools"))
    import bench_scenario0
    import oracle
    from dyobav_mpcnwta_warehouse_amd.solver import OptimizerSolution

    class OracleSolver:
        """solver().run(p) backed by the oracle; keeps the multipliers of the previous call like solver.Solver does"""

        def __init__(self, tracker, reassoc):
            c = tracker.solver.config                     # the configuration the HIP solver of this tracker was created with
            self.pr = oracle.Problem(c.N_hor, c.Nother, c.Nstcobs, c.Ndynobs, c.ts, c.lin_vel_min, c.lin_vel_max, c.ang_vel_max,
                                     c.lin_acc_min, c.lin_acc_max, c.ang_acc_max, c.vehicle_width, c.vehicle_margin, c.social_margin)
            self.op = oracle.Options(tolerance=c.tolerance, initial_tolerance=c.initial_tolerance, delta_tolerance=c.delta_tolerance,
                                     max_outer=c.max_outer_iterations, max_inner=c.max_inner_iterations, lbfgs_mem=c.lbfgs_memory,
                                     initial_penalty=c.initial_penalty, penalty_update=c.penalty_update_factor,
                                     inner_tol_update=c.inner_tolerance_update_factor, sufficient_decrease=c.sufficient_decrease_coeff,
                                     lip_delta=c.lip_delta_f64, lip_eps=c.lip_eps_f64, cbfgs_alpha=c.cbfgs_alpha,
                                     cbfgs_eps=c.cbfgs_epsilon, sy_eps=c.sy_epsilon, akkt_form=c.akkt_form, max_evals=c.max_evaluations, hoist_trig=1)
            self.reassoc = reassoc
            self.y = None
            self.n_converged = self.n_calls = 0
            tracker.solver.close() if hasattr(tracker.solver, "close") else None

        def run(self, p, initial_guess=None, initial_lagrange_multipliers=None, initial_penalty=None):
            u, y, res = oracle.solve(self.pr, self.op, np.asarray(p, dtype=np.float64).ravel(), u0=initial_guess,
                                     y0=self.y if initial_lagrange_multipliers is None else initial_lagrange_multipliers, reassoc=self.reassoc)
            self.y = y
            self.n_calls += 1
            self.n_converged += int(res["status"] == 0)
            st = {0: "Converged", 1: "NotConvergedIterations", 2: "NotConvergedOutOfTime"}.get(int(res["status"]), "NotFiniteComputation")
            return OptimizerSolution(exit_status=st, num_outer_iterations=int(res["outer_iters"]), num_inner_iterations=int(res["inner_iters"]),
                                     last_problem_norm_fpr=0.0, f1_infeasibility=0.0, f2_norm=0.0, solve_time_ms=0.0, penalty=0.0,
                                     solution=[float(v) for v in u], lagrange_multipliers=[float(v) for v in y], cost=float(res["cost"]))

    made = []

    def swap(reassoc):
        def f(tracker):
            made.append(OracleSolver(tracker, reassoc))
            return made[-1]
        return f

    os.environ["SCENARIO0_TRAJ"] = "1"
    try:
        r_hip = bench_scenario0.run(max_steps=160, with_map=True)
    finally:
        os.environ.pop("SCENARIO0_TRAJ")
    r_orc = bench_scenario0.run(max_steps=160, with_map=True, swap_solver=swap(False))
    r_twn = bench_scenario0.run(max_steps=160, with_map=True, swap_solver=swap(True))

    def dist(a, b):
        ta, tb = np.array(a["trajectory"]), np.array(b["trajectory"])
        n = min(len(ta), len(tb))
        return float(np.linalg.norm(ta[:n, :2] - tb[:n, :2], axis=1).max()), n
    d_hip, n_hip = dist(r_hip, r_orc)
    d_twn, n_twn = dist(r_twn, r_orc)
    print("scenario_0 + map: steps HIP %d / oracle %d / twin %d; largest distance between the robots along the run: HIP vs oracle %.3e m (%d steps), "
          "twin vs oracle %.3e m; oracle converged %d / %d solves" % (r_hip["steps"], r_orc["steps"], r_twn["steps"], d_hip, n_hip, d_twn,
                                                                       made[0].n_converged, made[0].n_calls))
    for r in (r_hip, r_orc, r_twn):
        assert r["reached_goal"] and r["min_pedestrian_distance_m"] > ev.HUMAN_SIZE, r
    assert abs(r_hip["steps"] - r_orc["steps"]) <= max(2, abs(r_twn["steps"] - r_orc["steps"]) + 1)
    # the kernels are no further from the oracle than its own re-association is (a benign scenario: most solves converge, the
    # loops stay together to within centimetres)
    assert d_hip <= max(3 * d_twn, 0.05), (d_hip, d_twn)
