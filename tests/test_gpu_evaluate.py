"""GPU test of the f3 path (batched closed-loop evaluator): B scenarios in lock-step on the device against the same
scenarios run one by one through the reference-API mirror (MpcInterface -> TrajectoryTracker -> solver().run, B = 1),
with the pedestrians and the constant-velocity predictor restated in numpy here (main_base.py:238-264, 293-302,
320-335; basic_agent.py:52-82; interfaces/cvmp_interface.py:44-57)."""
import types

import numpy as np
import pytest

import dyobav_mpcnwta_warehouse_amd as nm
from dyobav_mpcnwta_warehouse_amd.evaluate import HUMAN_SIZE, HUMAN_VMAX, BatchEvaluator
from dyobav_mpcnwta_warehouse_amd.mpc_interface import MpcInterface
from dyobav_mpcnwta_warehouse_amd.motion_model import unicycle_model

from dyobav_mpcnwta_warehouse_amd.solver import Solver

pytestmark = pytest.mark.gpu


def _cfg():
    # Lipschitz-estimator step 1e-4 on both sides: with OpEn's 1e-12 a 1e-12 difference in the inputs (torch vs numpy
    # arithmetic of the harness) already changes the first step length by ~1e-3 (parity protocol)
    cfg = nm.default_config_struct()
    cfg.lip_delta_f64 = cfg.lip_eps_f64 = 1e-4
    return cfg


def _scenarios(B, rng):
    boxes = []
    for i in range(14):
        c = np.array([1.5 + 1.1 * i, (-1) ** i * rng.uniform(1.6, 2.6)])
        hx, hy = rng.uniform(0.3, 0.6, 2)
        boxes.append([[c[0] + hx, c[1] + hy], [c[0] - hx, c[1] + hy], [c[0] - hx, c[1] - hy], [c[0] + hx, c[1] - hy]])
    starts = np.stack([np.zeros(B), rng.uniform(-0.4, 0.4, B), rng.uniform(-0.3, 0.3, B)], axis=1)
    paths = [[(float(8.0 + rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.5, 0.5)))] for _ in range(B)]
    hstart = np.stack([np.stack([rng.uniform(5, 9, B), rng.uniform(2.0, 3.5, B)], 1),
                       np.stack([rng.uniform(6, 10, B), rng.uniform(-3.5, -2.0, B)], 1)], axis=1)     # [B,2,2]
    hpath = np.stack([np.stack([hstart[:, 0] + np.array([-3.0, -5.0]), hstart[:, 0] + np.array([-6.0, -5.5])], 1),
                      np.stack([hstart[:, 1] + np.array([-2.5, 5.0]), hstart[:, 1] + np.array([-5.0, 5.5])], 1)], axis=1)
    return np.array(boxes), starts, paths, hstart, hpath


def _sequential(boxes, start, path, hstart, hpath, steps):
    geo = types.SimpleNamespace(processed_obstacle_list=[[tuple(v) for v in q] for q in boxes])
    mi = MpcInterface("mpc_fast.yaml", start.copy(), geo, verbose=False, solver_factory=lambda: Solver(_cfg()))
    mi.update_global_path(path)
    robot = start.copy()
    humans = [h.copy() for h in hstart]
    hist = [[h.copy()] for h in hstart]
    coming = [[tuple(w) for w in hp] for hp in hpath]
    traj, collision, complete, n = [robot.copy()], False, False, 0
    for _ in range(steps):
        rows = []
        for h, past in zip(humans, hist):
            pts = past[-5:]
            v = np.mean(np.diff(np.array(pts), axis=0), axis=0) if len(pts) > 1 else np.zeros(2)
            rows.append([[h[0], h[1], HUMAN_SIZE, HUMAN_SIZE, 0, 1]] +
                        [[h[0] + v[0] * (i + 1), h[1] + v[1] * (i + 1), 1.0, 1.0, 0, 1] for i in range(20)])
        mi.set_current_state(robot)
        actions, pred, cost, closest, refs = mi.run_step("work", rows, True)
        a = np.array(actions[0])
        if a[0] < 0:
            a = np.zeros(2)
        robot = unicycle_model(robot, a, 0.2)
        for i in range(len(humans)):
            if coming[i] and np.hypot(*(np.array(coming[i][0]) - humans[i])) < HUMAN_VMAX * 0.2:
                coming[i].pop(0)
            if coming[i]:
                d = np.array(coming[i][0]) - humans[i]
                humans[i] = humans[i] + 0.2 * d / np.hypot(*d) * HUMAN_VMAX
                hist[i].append(humans[i].copy())
        traj.append(robot.copy())
        n += 1
        dd = min(np.hypot(*(robot[:2] - h)) for h in humans)
        inside = any(min(q[:, 0]) < robot[0] < max(q[:, 0]) and min(q[:, 1]) < robot[1] < max(q[:, 1]) for q in boxes)
        if inside or dd <= HUMAN_SIZE:
            collision = True
            break
        if abs(robot[0] - path[-1][0]) <= 0.5 and abs(robot[1] - path[-1][1]) <= 0.5 and abs(a[0]) < 0.4:
            complete = True
            break
    return np.array(traj), collision, complete, n


class _ReplaySolver:
    """Stands where ``solver()`` stands in the tracker: checks the parameter vector the reference-API mirror built
    against the one the batched evaluator built for the same step, then answers with the batched solution, so the
    tracker's internal state (previous action, reference index) follows the batched loop exactly."""

    def __init__(self, record, b):
        self.record, self.b, self.k, self.worst = record, b, 0, 0.0

    def run(self, p, *a, **kw):
        rec = self.record[self.k]
        self.worst = max(self.worst, float(np.abs(np.asarray(p) - rec["P"][self.b]).max()))
        self.k += 1
        return types.SimpleNamespace(solution=rec["U"][self.b].tolist(), cost=0.0, exit_status="Converged",
                                     solve_time_ms=0.0)


def test_every_step_of_the_batched_loop_is_the_step_the_reference_api_takes():
    """Teacher-forced equivalence: the early solves of a run stop at the iteration caps (accelerating from rest keeps
    the ALM constraints active) and amplify a 1-ulp difference of the state by ~1e16, so two free-running loops
    separate by ~1e-3 after two steps whatever the implementation. Step by step on the SAME history instead:
    (1) the sequential API builds the same parameter vector, (2) the solver gives the same answer for it at B = 1,
    (3) the agents move as the reference's models move them."""
    rng = np.random.default_rng(12)
    B, steps = 5, 25
    boxes, starts, paths, hstart, hpath = _scenarios(B, rng)
    ev = BatchEvaluator(_cfg(), starts, paths, hstart, hpath, boxes, dtype=np.float64)
    record = []
    res = ev.run(max_steps=steps, record=record)
    ev.close()
    assert len(record) == len(res.solve_ms) >= 10
    geo = types.SimpleNamespace(processed_obstacle_list=[[tuple(v) for v in q] for q in boxes])
    h = nm.Handle(_cfg())
    for b in range(B):
        n = int(res.steps[b])
        replay = _ReplaySolver(record, b)
        mi = MpcInterface("mpc_fast.yaml", starts[b].copy(), geo, verbose=False, solver_factory=lambda: replay)
        mi.update_global_path(paths[b])
        hist = [[] for _ in range(hstart.shape[1])]
        for k in range(n):
            rec = record[k]
            assert rec["alive"][b]
            rows = []
            for i, past in enumerate(hist):
                hpos = rec["humans"][b, i]
                if not past or not np.array_equal(past[-1], hpos):       # past_traj grows only while moving
                    past.append(hpos.copy())
                pts = past[-5:]
                v = np.mean(np.diff(np.array(pts), axis=0), axis=0) if len(pts) > 1 else np.zeros(2)
                rows.append([[hpos[0], hpos[1], HUMAN_SIZE, HUMAN_SIZE, 0, 1]] +
                            [[hpos[0] + v[0] * (j + 1), hpos[1] + v[1] * (j + 1), 1.0, 1.0, 0, 1] for j in range(20)])
            mi.set_current_state(rec["robot"][b].copy())
            mi.run_step("work", rows, True)
            # (2) same problem, same multipliers, alone in a batch of one -> same answer, bit for bit
            if k in (0, 1, n // 2, n - 1):
                o = h.solve(rec["P"][b:b + 1], y0=rec["y_in"][b:b + 1], dtype=np.float64)
                assert np.array_equal(o["U"][0], rec["U"][b]), (b, k)
            # (3) robot motion = the reference's motion model on the clipped first action
            a = rec["U"][b, :2].copy()
            if a[0] < 0:
                a[:] = 0
            nxt = unicycle_model(rec["robot"][b], a, 0.2)
            assert np.abs(res.trajectory[b, k + 1] - nxt).max() < 1e-13, (b, k)
            if k + 1 < n:
                assert np.abs(record[k + 1]["robot"][b] - nxt).max() < 1e-13
                for i in range(hstart.shape[1]):                       # pedestrians: basic_agent.py:52-82
                    hpos, left = rec["humans"][b, i], [w for w in hpath[b, i]]
                    # way-points already passed = those the recorded positions have come within one step of
                    passed = 0
                    for kk in range(k + 1):
                        if passed < len(left) and np.hypot(*(left[passed] - record[kk]["humans"][b, i])) < HUMAN_VMAX * 0.2:
                            passed += 1
                    exp = hpos
                    if passed < len(left):
                        d = left[passed] - hpos
                        exp = hpos + 0.2 * d / np.hypot(*d) * HUMAN_VMAX
                    assert np.abs(record[k + 1]["humans"][b, i] - exp).max() < 1e-13, (b, k, i)
        assert replay.k == n
        assert replay.worst < 1e-12, (b, replay.worst)               # (1) parameter vectors of every step
    h.close()


def test_free_running_batched_and_sequential_loops_stay_together():
    """The two loops run independently (no teacher forcing): same outcome flags, trajectories within the spread the
    cap-limited early solves allow (see the test above for why not tighter)."""
    rng = np.random.default_rng(12)
    B, steps = 5, 30
    boxes, starts, paths, hstart, hpath = _scenarios(B, rng)
    ev = BatchEvaluator(_cfg(), starts, paths, hstart, hpath, boxes, dtype=np.float64)
    res = ev.run(max_steps=steps)
    ev.close()
    assert len(res.solve_ms) <= steps and res.trajectory.shape[0] == B
    same_flags, spread = 0, []
    for b in range(B):
        traj, col, comp, n = _sequential(boxes, starts[b], paths[b], hstart[b], hpath[b], steps)
        assert np.abs(res.trajectory[b, :2] - traj[:2]).max() < 1e-15, b      # first solve: identical inputs
        same_flags += int(res.steps[b] == n and bool(res.complete[b]) == comp and
                          bool(res.collision[b]) == (col or not comp))    # a time-out is booked as a collision
        m = min(n, int(res.steps[b]))
        spread.append(float(np.abs(res.trajectory[b, :m + 1] - traj[:m + 1]).max()))
    print("free-running spread per scenario:", spread, "same flags:", same_flags)
    assert same_flags >= B - 1 and np.median(spread) < 5e-2
    assert np.isfinite(res.deviation).all() and (res.clearance >= 0).all() and (res.clearance_dyn > 0).all()


def test_large_batch_runs_and_reports_metrics():
    rng = np.random.default_rng(13)
    B = 512
    boxes, starts, paths, hstart, hpath = _scenarios(B, rng)
    ev = BatchEvaluator(nm.default_config_struct(), starts, paths, hstart, hpath, boxes, dtype=np.float32,
                        human_stagger=0.2, seed=5)
    res = ev.run(max_steps=60)
    ev.close()
    assert res.complete.sum() + res.collision.sum() == B
    assert res.complete.mean() > 0.3                                     # most robots reach the goal region
    ok = res.complete
    # structural facts for every completed run; the deviation bound is statistical (the early, cap-limited solves
    # make single trajectories sensitive to rounding, see the teacher-forced test above)
    assert np.isfinite(res.smoothness[ok]).all() and np.quantile(res.deviation[ok, 0], 0.95) < 1.5
    assert (res.steps[ok] <= 60).all() and (res.clearance_dyn[ok] > HUMAN_SIZE).all()


def test_compaction_of_finished_scenarios_changes_nothing():
    """Large batches solve only the scenarios still running; same kernel family, so the results are identical."""
    rng = np.random.default_rng(14)
    B = 300
    boxes, starts, paths, hstart, hpath = _scenarios(B, rng)
    out = []
    for compact in (False, True):
        cfg = nm.default_config_struct()
        cfg.latency_waves = 2                    # pin the kernel: the automatic choice depends on the batch size
        ev = BatchEvaluator(cfg, starts, paths, hstart, hpath, boxes, dtype=np.float32, human_stagger=0.2, seed=5,
                            compact=compact)
        out.append(ev.run(max_steps=55))
        ev.close()
    a, b = out
    assert a.complete.sum() > 50 and (a.steps < 55).any()                 # scenarios did finish along the way
    for key in ("collision", "complete", "steps", "trajectory", "clearance", "clearance_dyn", "deviation"):
        assert np.array_equal(getattr(a, key), getattr(b, key), equal_nan=True), key
    assert np.array_equal(a.actions, b.actions, equal_nan=True)


@pytest.mark.parametrize("compact", [False, True])
def test_fused_step_kernels_against_the_torch_expressions(compact):
    """The two HIP kernels around the solve (nmpc_loop_pre / nmpc_loop_post, csrc/nmpc_step.h) against the same time step
    written out as torch expressions (``fused=False``, the implementation of rounds 1-2): what the first steps feed the
    solver and produce agrees to rounding (fp64), every flag and counter of a whole run is the same, metrics to 1e-9 --
    with stagger replayed so that both sides see the same pedestrians."""
    import torch
    rng = np.random.default_rng(21)
    B = 96
    boxes, starts, paths, hstart, hpath = _scenarios(B, rng)
    H = hstart.shape[1]
    draws = [torch.from_numpy(rng.integers(-10, 11, (B, H)) / 10 * 0.2).cuda() for _ in range(40)]
    out, recs = [], []
    for fused in (False, True):
        cfg = nm.default_config_struct()
        cfg.latency_waves = 2
        ev = BatchEvaluator(cfg, starts, paths, hstart, hpath, boxes, dtype=np.float64, compact=compact, fused=fused)
        ev.stagger_replay = [d.clone() for d in draws]
        rec = []
        out.append(ev.run(max_steps=40, record=rec))
        recs.append(rec)
        ev.close()
    a, b = out
    assert len(recs[0]) == len(recs[1]) >= 20
    # step 0: identical solver inputs, hence identical controls and identical states after it
    for key in ("robot", "humans", "P", "U"):
        x, y = recs[0][0][key], recs[1][0][key]
        assert np.abs(x - y).max() <= 1e-13 * max(1.0, np.abs(x).max()), key
    assert np.abs(a.trajectory[:, 1] - b.trajectory[:, 1]).max() < 1e-12
    assert np.abs(recs[0][1]["humans"] - recs[1][1]["humans"]).max() < 1e-12
    # later steps: the inputs agree to rounding; a solve that stops at its iteration caps amplifies a last-bit difference
    # of its input (see the teacher-forced test above), so the controls are compared in the median, the runs statistically
    live = recs[0][1]["alive"] & recs[1][1]["alive"]
    dP = np.abs(recs[0][1]["P"] - recs[1][1]["P"])[live].max(axis=1)
    assert dP.max() < 1e-12, dP.max()            # (measured 9e-15; the controls of such solves differ by up to 0.4)
    assert abs(a.complete.mean() - b.complete.mean()) <= 0.1 and abs(a.steps.mean() - b.steps.mean()) <= 3.0
    same = (a.steps == b.steps) & (a.complete == b.complete) & (a.collision == b.collision)
    assert same.mean() > 0.5
    # (rounding-level input differences grow along a closed loop of cap-limited solves: ~1e-3 m over a run)
    assert np.median(np.abs(a.clearance_dyn - b.clearance_dyn)[same]) < 5e-2
    assert np.median(np.abs(a.deviation - b.deviation)[same]) < 5e-2


def test_loop_kernels_refuse_host_pointers():
    """nmpc_loop_pre / nmpc_loop_post take device pointers only and say so instead of faulting."""
    import ctypes as C
    from dyobav_mpcnwta_warehouse_amd import _capi
    with nm.Handle(nm.default_config_struct()) as h:
        a = _capi.NmpcLoopArgs()
        host = np.zeros(4096)
        a.B = a.n_run = a.H = a.W = a.Lmax = 1
        a.max_steps, a.step, a.M = 4, 0, 0
        for name, _ in _capi.NmpcLoopArgs._fields_:
            if name not in ("B", "n_run", "H", "W", "Lmax", "M", "step", "max_steps", "run", "base_speed", "lin_vel_max",
                            "human_size", "human_vmax", "gather_y", "n_hyp", "hyp_fan_rad", "hyp_radius0", "hyp_radius_growth", "polys",
                            "stagger"):
                setattr(a, name, host.ctypes.data)
        with pytest.raises(nm.NmpcError, match="device pointer"):
            h.loop_step(np.float64, a, post=False)
        a.robot = None
        with pytest.raises(nm.NmpcError, match="NULL"):
            h.loop_step(np.float64, a, post=True)
