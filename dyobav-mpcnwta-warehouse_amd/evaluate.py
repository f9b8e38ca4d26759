"""Batched closed-loop evaluator ("next" row f3): B independent warehouse scenarios advanced in lock-step on the device.

Semantics of the reference's evaluation loop for the MPC tracker with the constant-velocity predictor
(``/root/reference/src``):

* ``MainBase.run_once`` / ``run_one_step``                      main_base.py:267-346, 348-425
* ``MainBase.run_cv_prediction`` + ``CvmpInterface``            main_base.py:238-264, interfaces/cvmp_interface.py:24-57
  (mean step of the last <= 5 positions, extrapolated; std 1.0 for predicted offsets, HUMAN_SIZE at offset 0)
* obstacle rows ``[mu_x, mu_y, std_x, std_y, 0, 1]``            main_base.py:293-302
* ``MpcInterface.run_step`` -> ``TrajectoryTracker.run_step``   interfaces/mpc_interface.py:52-71, trajectory_tracker.py:273-337
  (reference-state window, speed-reference rule, previous action; multipliers carried between solves)
* no-backward clip, robot / pedestrian motion                  main_base.py:320-324, basic_agent.py:52-82
* metrics                                                       main_pre.py:20-53, main_base.py:326-335, 427-435

One time step = CV prediction -> obstacle rows -> reference windows -> ``nmpc_assemble_params`` (f1) ->
``nmpc_solve_batch`` -> first action -> agent motion -> metrics, for all scenarios at once; nothing leaves HBM
between steps. Everything around the solve is two HIP kernels (``nmpc_loop_pre_*`` / ``nmpc_loop_post_*``,
``csrc/nmpc_step.h``); the torch expressions further down (``fused=False``) are the same arithmetic written out op by
op -- they were the implementation of rounds 1-2 and stay as the independent check of the kernels. Where the reference runs ``max_num_run`` scenarios one after another (main_base.py:448-464), this
runs them side by side. Pedestrian stagger uses a seeded torch generator (the reference's ``random`` is unseeded).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import _capi
from .trajectory_tracker import TrajectoryTracker

HUMAN_SIZE = 0.2   # main_base.py:75, main_pre.py:18
HUMAN_VMAX = 1.5   # main_base.py:76


@dataclass
class EvaluationResult:
    collision: np.ndarray        # [B] bool   (main_base.py:366-371; a time-out counts as a collision, :407-410)
    complete: np.ndarray         # [B] bool
    steps: np.ndarray            # [B] int    time steps executed
    smoothness: np.ndarray       # [B, 2]     mean |2nd difference| of (v, w)        (main_pre.calc_action_smoothness)
    clearance: np.ndarray        # [B]        min distance to the static polygons     (calc_minimal_obstacle_distance)
    clearance_dyn: np.ndarray    # [B]        min distance to a pedestrian            (calc_minimal_dynamic_obstacle_distance)
    deviation: np.ndarray        # [B, 2]     mean / max distance to the reference trajectory (calc_deviation_distance)
    trajectory: np.ndarray       # [B, T+1, 3] robot states (rows after the end repeat the last state)
    actions: np.ndarray          # [B, T, 2]  raw solver actions
    solve_ms: List[float]        # kernel time of every batched solve


def action_smoothness(A):
    """``main_pre.calc_action_smoothness`` (main_pre.py:34-37) for a batch: mean |second difference| of (v, w) over the
    actions taken; ``A`` [B, T, 2] torch tensor, NaN rows = steps after the run ended. Returns [B, 2] (NaN for T < 3)."""
    import torch
    B = A.shape[0]
    smooth = torch.full((B, 2), float("nan"), dtype=A.dtype, device=A.device)
    if A.shape[1] >= 3:
        d2 = (A[:, 2:] - 2 * A[:, 1:-1] + A[:, :-2]).abs()
        valid = ~torch.isnan(d2[..., 0])
        smooth = torch.nan_to_num(d2, nan=0.0).sum(dim=1) / valid.sum(dim=1).clamp(min=1)[:, None].to(A.dtype)
    return smooth


def min_dynamic_distance(robot_xy, humans):
    """``main_pre.calc_minimal_dynamic_obstacle_distance`` (main_pre.py:45-47): [B] distance from the robot to its
    closest pedestrian; ``robot_xy`` [B, 2], ``humans`` [B, H, 2]."""
    import torch
    return torch.linalg.norm(robot_xy[:, None, :] - humans, dim=-1).min(dim=1).values


def deviation_to_reference(robot_xy, ref_traj, ref_len):
    """One term of ``main_pre.calc_deviation_distance`` (main_pre.py:49-53): [B] distance from the robot position to
    the closest point of its reference trajectory (``ref_traj`` [B, L, 2+], first ``ref_len[b]`` rows valid)."""
    import torch
    d = torch.cdist(robot_xy[:, None, :], ref_traj[:, :, :2])[:, 0]
    inf = torch.full_like(d, float("inf"))
    return torch.where(torch.arange(d.shape[1], device=d.device)[None] < ref_len[:, None], d, inf).min(dim=1).values


def unicycle_rk4_step(robot, act, ts):
    """``basic_agent.Robot.one_step`` = ``UnicycleModel(ts, rk4=True)`` (basic_motion_model/motion_model.py:141-163) in
    closed form for a batch: ``robot`` [B, 3], ``act`` [B, 2] -> [B, 3]."""
    import torch
    th, v, w = robot[:, 2], act[:, 0], act[:, 1]
    hh = 0.5 * ts * w
    cc = (torch.cos(th) + 4 * torch.cos(th + hh) + torch.cos(th + 2 * hh)) / 6
    ss = (torch.sin(th) + 4 * torch.sin(th + hh) + torch.sin(th + 2 * hh)) / 6
    return torch.stack([robot[:, 0] + ts * v * cc, robot[:, 1] + ts * v * ss, th + ts * w], dim=1)


class BatchEvaluator:
    def __init__(self, config: _capi.NmpcConfigStruct, robot_starts: np.ndarray, robot_paths: Sequence[Sequence[tuple]],
                 human_starts: np.ndarray, human_paths: np.ndarray, map_polygons: np.ndarray, dtype=np.float64,
                 human_stagger: float = 0.0, seed: int = 0, mode: str = "work",
                 tuning: Optional[Sequence[float]] = None, lin_vel_max: float = 1.5, warm_start: bool = False,
                 compact: Optional[bool] = None, fused: bool = True, n_hyp: int = 1, hyp_fan: float = 0.15,
                 hyp_radius_growth: float = 0.05):
        """``n_hyp`` > 1: every pedestrian enters the solver as ``n_hyp`` obstacle rows fanned around its constant-velocity
        prediction by ``(j - (n_hyp - 1) / 2) * hyp_fan`` rad, radii ``HUMAN_SIZE + hyp_radius_growth * t`` -- the
        multi-hypothesis obstacle tensor SURVEY.md 8(d) prescribes for BASELINE configs[2] (4 pedestrians x 10
        hypotheses), here produced closed-loop from the scenarios' own pedestrian motion instead of one-shot. ``n_hyp``
        = 1 is the reference's constant-velocity predictor (one row per pedestrian, std 1.0)."""
        import torch
        self.torch = torch
        self.fused = fused
        self.time_solves = True     # record the HIP-event time of every batched solve (one event wait per time step)
        self.cfg = config
        self.dt = np.dtype(dtype)
        self.tdt = torch.float32 if self.dt == np.float32 else torch.float64
        self.dev = torch.device("cuda", config.device_id)
        if config.axis_aligned == 0:
            # the pedestrians' obstacle rows are written here with angle = 0 (main_base.py:302), so the promise can be
            # made once instead of leaving the code path to the per-call scan of the batch (nmpc_hip.h, axis_aligned:
            # with 0 the path -- and with it the last bits of every result -- depends on the batch composition)
            import copy
            config = copy.copy(config)
            config.axis_aligned = 1
            self.cfg = config
        li = _capi.layout_info(config)
        streamed = bool(li.global_table_f32 if self.dt == np.float32 else li.global_table_f64)
        # (obstacle table streamed from global memory -- e.g. N = 40, 160 rows: the library's automatic choice there is
        #  the cooperative kernel, whatever the batch size; pinning latency_waves would switch it off)
        if config.latency_waves == 0 and not streamed and (robot_starts.shape[0] > 1024 if compact is None else compact):
            # The batch shrinks as scenarios finish (compaction) and the library's automatic choice between its kernel
            # families follows the batch size. A scenario's closed-loop trajectory must not depend on who else is still
            # running, so either every plan computes the same bits -- fp32 with the obstacle table in registers:
            # nmpc_config.batch_invariant, and the library goes on choosing the fastest plan for what is left (latency
            # kernels for the last few hundred scenarios) -- or, where the families agree to rounding only (fp64, LDS
            # table), the family is fixed here from the initial batch.
            import copy
            config = copy.copy(config)
            if self.dt == np.float32 and li.reg_slots_f32 > 0:
                config.batch_invariant = 1
            else:
                n_simd = 4 * self.torch.cuda.get_device_properties(self.dev).multi_processor_count
                config.latency_waves = 1 if robot_starts.shape[0] > 4 * n_simd else 2
            self.cfg = config
        self.h = _capi.Handle(config)
        self.h.set_stream(torch.cuda.current_stream(self.dev).cuda_stream)
        # Within the latency-kernel family the number of wavefronts per instance does not change the result (the W
        # candidates of a round are evaluated with the same arithmetic as one after the other; bit-identical for W = 2,
        # 3, 4 -- tests/test_gpu_options.py), so W follows the number of scenarios still running: as many wavefronts per
        # instance as stay resident together. One handle per W.
        self._h_by_waves = {}
        if config.latency_waves >= 2:
            import copy
            self._n_simd = 4 * torch.cuda.get_device_properties(self.dev).multi_processor_count
            for w in (2, 3, 4):
                if w == config.latency_waves:
                    self._h_by_waves[w] = self.h
                else:
                    c = copy.copy(config)
                    c.latency_waves = w
                    self._h_by_waves[w] = _capi.Handle(c)
                    self._h_by_waves[w].set_stream(torch.cuda.current_stream(self.dev).cuda_stream)
        self.N, self.ts = config.N_hor, config.ts
        B = self.B = robot_starts.shape[0]
        T = lambda x, dt=None: torch.as_tensor(np.ascontiguousarray(x), dtype=dt or self.tdt, device=self.dev)
        self.robot = T(robot_starts)                               # [B,3]
        self.goal = T(np.array([p[-1] for p in robot_paths], dtype=float))   # [B,2]
        self.humans = T(human_starts)                              # [B,H,2]
        self.H = self.humans.shape[1]
        self.n_hyp, self.hyp_fan, self.hyp_grow = max(1, int(n_hyp)), float(hyp_fan), float(hyp_radius_growth)
        if self.H * self.n_hyp > config.Ndynobs:
            raise ValueError(f"{self.H} pedestrians x {self.n_hyp} hypotheses exceed Ndynobs = {config.Ndynobs}")
        # hook for harvesting (scenarios.harvest_closed_loop): called as on_params(kt, idx, Pa) after the parameter
        # vectors of time step kt have been assembled (idx = running scenarios of the compact batch, None = all)
        self.on_params = None
        self.hpath = T(human_paths)                                # [B,H,W,2]
        self.hidx = torch.zeros(B, self.H, dtype=torch.long, device=self.dev)
        self.hist = self.humans[:, :, None, :].repeat(1, 1, 5, 1)  # last <= 5 positions, newest last
        self.hcount = torch.ones(B, self.H, dtype=torch.long, device=self.dev)
        self.polys = T(map_polygons)                               # [M,4,2]
        self.stagger = float(human_stagger)
        self.warm_start = warm_start       # row f4 (extension): shifted previous solution as the initial guess
        self.compact = (robot_starts.shape[0] > 1024) if compact is None else bool(compact)
        self.gen = torch.Generator(device=self.dev).manual_seed(seed)
        # replay hook for tests: a list of [B, H] tensors, one per pedestrian step, used instead of the generator
        # (the reference draws the stagger with python's `random`, basic_agent.py:64; a recording can be replayed)
        self.stagger_replay: Optional[list] = None
        scale = {"safe": 0.2, "work": 0.8, "super": 1.0}[mode]
        self.base_speed = lin_vel_max * scale
        self.lin_vel_max = lin_vel_max
        # global reference trajectories: TrajectoryTracker.get_ref_traj per scenario (host, once), padded
        # (the Monte-Carlo runs of one scenario share start and path: make_reference_scenarios has three distinct pairs
        #  for any B -- one host call per distinct pair)
        cache = {}

        def ref_traj(p, s):
            key = (tuple(map(tuple, p)), tuple(float(v) for v in s))
            if key not in cache:
                cache[key] = np.array(TrajectoryTracker.get_ref_traj(self.ts, list(p), tuple(s), self.base_speed))
            return cache[key]
        trajs = [ref_traj(p, s) for p, s in zip(robot_paths, robot_starts)]
        self.ref_len = torch.as_tensor([len(t) for t in trajs], device=self.dev)
        Lmax = max(len(t) for t in trajs)
        pad = np.stack([np.concatenate([t, np.repeat(t[-1:], Lmax - len(t), axis=0)]) for t in trajs])
        self.ref_traj = T(pad)                                     # [B,Lmax,3]
        self.idx_ref = torch.zeros(B, dtype=torch.long, device=self.dev)
        self.tuning = T(np.asarray(tuning if tuning is not None else
                                   (0.0, 10.0, 0.0, 0.0, 0.0, 0.0, 0.0, 100.0, 10.0, 20.0), dtype=float))
        self.stcw = torch.full((self.N,), 10.0, dtype=self.tdt, device=self.dev)   # set_obstacle_weights(10, 10)
        self.dynw = torch.full((self.N,), 10.0, dtype=self.tdt, device=self.dev)
        self.P = torch.empty(B, self.h.np_, dtype=self.tdt, device=self.dev)
        self.U = torch.empty(B, 2 * self.N, dtype=self.tdt, device=self.dev)
        self._Ua = torch.empty(B, 2 * self.N, dtype=self.tdt, device=self.dev)   # compacted batch of the running scenarios
        # dispatch order of a time step's solves from the evaluation counts of the previous one (nmpc_set_dispatch_order)
        self.dispatch_by_history, self.dispatch_min_batch = True, 32768   # (measured: -4 % at 16 384 scenarios, +10 % at 65 536)
        self._info = torch.zeros(B, 8, dtype=self.tdt, device=self.dev)
        self._evals = torch.zeros(B, dtype=self.tdt, device=self.dev)
        self.y = torch.zeros(B, 2 * self.N, dtype=self.tdt, device=self.dev)
        self.status = torch.empty(B, dtype=torch.int32, device=self.dev)
        # count_status: the exit statuses of every time step's solves are counted on the device (status_counts: one [4]
        # tensor per step -- Converged / NotConvergedIterations / NotConvergedOutOfTime / anything else) -- how often the
        # evaluation budget (nmpc_config.max_evaluations) cuts a solve off, and how often the loop runs on a converged answer
        self.count_status = False
        self.status_counts: list = []

    # ---------------------------------------------------------------------------------------------------------
    def _predict_cv(self):
        """[B,H,N+1,6] obstacle rows from the constant-velocity extrapolation of the last <= 5 positions."""
        torch = self.torch
        B, H, N = self.B, self.H, self.N
        diffs = self.hist[:, :, 1:, :] - self.hist[:, :, :-1, :]                 # [B,H,4,2], newest last
        nd = (self.hcount - 1).clamp(min=0, max=4)                               # usable differences
        k = torch.arange(4, device=self.dev)[None, None, :]
        mask = (k >= (4 - nd[..., None])).to(self.tdt)[..., None]
        vel = (diffs * mask).sum(dim=2) / nd.clamp(min=1)[..., None].to(self.tdt)
        off = torch.arange(0, N + 1, device=self.dev, dtype=self.tdt)[None, None, :, None]
        if self.n_hyp > 1:       # hypothesis fan around the constant-velocity step (nmpc_hip.h, nmpc_loop_args::n_hyp)
            nh = self.n_hyp
            ang = (torch.arange(nh, device=self.dev, dtype=self.tdt) - 0.5 * (nh - 1)) * self.hyp_fan
            ca, sa = torch.cos(ang)[None, None, :], torch.sin(ang)[None, None, :]
            wx = ca * vel[:, :, None, 0] - sa * vel[:, :, None, 1]                 # [B,H,nh]
            wy = sa * vel[:, :, None, 0] + ca * vel[:, :, None, 1]
            w = torch.stack([wx, wy], dim=-1).reshape(B, H * nh, 2)
            cur = self.humans[:, :, None, :].expand(-1, -1, nh, -1).reshape(B, H * nh, 2)
            rows = torch.zeros(B, H * nh, N + 1, 6, dtype=self.tdt, device=self.dev)
            rows[..., 0:2] = cur[:, :, None, :] + w[:, :, None, :] * off
            rows[..., 2:4] = (HUMAN_SIZE + self.hyp_grow * off)
            rows[..., 5] = 1.0
            return rows
        rows = torch.zeros(B, H, N + 1, 6, dtype=self.tdt, device=self.dev)
        rows[..., 0:2] = self.humans[:, :, None, :] + vel[:, :, None, :] * off
        rows[..., 2:4] = 1.0
        rows[:, :, 0, 2:4] = HUMAN_SIZE
        rows[..., 5] = 1.0
        return rows

    def _ref_states(self):
        """TrajectoryTracker.get_ref_states for all scenarios (note the reference passes N_hor as ``action_steps``,
        trajectory_tracker.py:186-187, so the search window is [idx - N, idx + 5N); N rows are taken -- the
        reference's default ``horizon=20`` equals N_hor in both shipped yaml files)."""
        torch = self.torch
        N, hor = self.N, self.N
        Lmax = self.ref_traj.shape[1]
        j = torch.arange(Lmax, device=self.dev)[None, :]
        lo = (self.idx_ref - N).clamp(min=0)[:, None]
        hi = torch.minimum(self.ref_len, self.idx_ref + 5 * N)[:, None]
        d = torch.hypot(self.robot[:, None, 0] - self.ref_traj[:, :, 0], self.robot[:, None, 1] - self.ref_traj[:, :, 1])
        d = torch.where((j >= lo) & (j < hi), d, torch.full_like(d, float("inf")))
        self.idx_ref = torch.argmin(d, dim=1)               # first minimum, like list.index(min(...))
        rows = self.idx_ref[:, None] + torch.arange(hor, device=self.dev)[None, :]
        rows = torch.minimum(rows, (self.ref_len - 1)[:, None])
        return torch.gather(self.ref_traj, 1, rows[..., None].expand(-1, -1, 3)).contiguous()[:, :N]

    def _in_polygon(self, pts):
        """[B] strictly inside any convex quadrilateral (shapely Polygon.contains)."""
        a = self.polys[None]                                                     # [1,M,4,2]
        b = self.torch.roll(self.polys, -1, dims=1)[None]
        p = pts[:, None, None, :]
        cross = (b[..., 0] - a[..., 0]) * (p[..., 1] - a[..., 1]) - (b[..., 1] - a[..., 1]) * (p[..., 0] - a[..., 0])
        inside = (cross > 0).all(dim=2) | (cross < 0).all(dim=2)
        return inside.any(dim=1)

    def _polygon_distance(self, pts):
        """[B] distance to the closest polygon (0 inside), shapely Polygon.distance(Point)."""
        torch = self.torch
        a = self.polys[None]
        b = torch.roll(self.polys, -1, dims=1)[None]
        p = pts[:, None, None, :]
        ab = b - a
        t = (((p - a) * ab).sum(-1) / (ab * ab).sum(-1)).clamp(0, 1)
        d = torch.linalg.norm(a + t[..., None] * ab - p, dim=-1).min(dim=2).values     # [B,M]
        cross = ab[..., 0] * (p[..., 1] - a[..., 1]) - ab[..., 1] * (p[..., 0] - a[..., 0])
        inside = (cross > 0).all(dim=2) | (cross < 0).all(dim=2)
        return torch.where(inside, torch.zeros_like(d), d).min(dim=1).values

    def _step_humans(self):
        torch = self.torch
        W = self.hpath.shape[2]
        tgt = torch.gather(self.hpath, 2, self.hidx.clamp(max=W - 1)[..., None, None].expand(-1, -1, 1, 2))[:, :, 0]
        dist = torch.linalg.norm(tgt - self.humans, dim=-1)
        adv = (dist < HUMAN_VMAX * self.ts) & (self.hidx < W)                    # basic_agent.py:57-59
        self.hidx = self.hidx + adv.long()
        moving = self.hidx < W
        tgt = torch.gather(self.hpath, 2, self.hidx.clamp(max=W - 1)[..., None, None].expand(-1, -1, 1, 2))[:, :, 0]
        dist = torch.linalg.norm(tgt - self.humans, dim=-1).clamp(min=1e-12)
        dire = (tgt - self.humans) / dist[..., None]
        if self.stagger_replay is not None:
            st = self.stagger_replay.pop(0).to(self.tdt)[..., None]
        elif self.stagger > 0:
            sign = torch.randint(0, 2, dist.shape, generator=self.gen, device=self.dev) * 2 - 1
            mag = torch.randint(0, 11, dist.shape, generator=self.gen, device=self.dev).to(self.tdt) / 10
            st = (sign.to(self.tdt) * mag * self.stagger)[..., None]
        else:
            st = 0.0
        new = self.humans + self.ts * (dire * HUMAN_VMAX + st)                   # omnidirectional model
        self.humans = torch.where(moving[..., None], new, self.humans)
        # past_traj grows only while the pedestrian moves (basic_agent.py:72-82)
        shifted = torch.cat([self.hist[:, :, 1:], self.humans[:, :, None, :]], dim=2)
        self.hist = torch.where(moving[..., None, None], shifted, self.hist)
        self.hcount = self.hcount + moving.long()

    # ---------------------------------------------------------------------------------------------------------
    def _solve(self, hs, Pa, nA, Ua, u0, ya, y_is_input, info, status=None):
        """The MPC solves of one time step: parameters ``Pa[nA, np]`` -> controls ``Ua[nA, 2N]``, multipliers ``ya`` in / out
        (device tensors; main_base.py:308-311 per scenario), optionally the exit statuses. One overridable call, so that a
        checker can drive the same closed loop with another solver (tests/test_gpu_closed_loop.py puts the CPU oracle here)."""
        hs.solve_raw(self.dt, Pa, nA, Ua, status=status, u0=u0, y=ya, y_is_input=y_is_input, info=info, sync=False)

    def run(self, max_steps: int = 120, record: Optional[list] = None) -> EvaluationResult:
        """``record``: if a list, one dict per time step is appended with host copies of what the step saw and
        produced (robot, humans, P, y_in, U) -- for step-by-step checks against the sequential API; costs a
        device->host copy per step."""
        return self._run_fused(max_steps, record) if self.fused else self._run_torch(max_steps, record)

    def _run_fused(self, max_steps, record):
        """The time step as two HIP kernels around f1 + solve (csrc/nmpc_step.h)."""
        torch = self.torch
        B, N, H = self.B, self.N, self.H
        dev, tdt = self.dev, self.tdt
        z = lambda *shape, dtype=tdt, fill=0.0: torch.full(shape, fill, dtype=dtype, device=dev)
        alive = z(B, dtype=torch.uint8, fill=1)
        collision, complete = z(B, dtype=torch.uint8, fill=0), z(B, dtype=torch.uint8, fill=0)
        steps = z(B, dtype=torch.long, fill=0)
        last_u = z(B, 2)
        traj = torch.empty(B, max_steps + 1, 3, dtype=tdt, device=dev)
        traj[:, 0] = self.robot
        acts = z(B, max(max_steps, 1), 2, fill=float("nan"))
        clr_dyn = z(B, fill=float("inf"))
        clr_stc = self._polygon_distance(self.robot[:, :2]).contiguous()
        d0 = torch.cdist(self.robot[:, None, :2], self.ref_traj[:, :, :2])[:, 0]
        dev_sum = d0.min(dim=1).values.clone()      # trajectory metrics include the start state (robot.past_traj[0])
        dev_max = dev_sum.clone()
        n_traj = z(B, fill=1.0)
        state_c, last_u_c, refs_c, speed_c = z(B, 3), z(B, 2), z(B, N, 3), z(B)
        dyn_c = z(B, H * self.n_hyp, N + 1, 6)
        ya_buf = z(B, 2 * N)
        self.hidx = self.hidx.contiguous()
        self.hcount = self.hcount.contiguous()
        self.idx_ref = self.idx_ref.contiguous()
        self.hist = self.hist.contiguous()
        self.humans = self.humans.contiguous()
        self.robot = self.robot.contiguous()
        ref_len = self.ref_len.to(torch.long).contiguous()
        a = _capi.NmpcLoopArgs()
        a.B, a.H, a.W, a.Lmax, a.M, a.max_steps = B, H, int(self.hpath.shape[2]), int(self.ref_traj.shape[1]), int(self.polys.shape[0]), max(max_steps, 1)
        a.base_speed, a.lin_vel_max, a.human_size, a.human_vmax = self.base_speed, self.lin_vel_max, HUMAN_SIZE, HUMAN_VMAX
        a.n_hyp, a.hyp_fan_rad, a.hyp_radius0, a.hyp_radius_growth = self.n_hyp, self.hyp_fan, HUMAN_SIZE, self.hyp_grow
        for name, t in (("robot", self.robot), ("last_u", last_u), ("humans", self.humans), ("hist", self.hist), ("hcount", self.hcount),
                        ("hidx", self.hidx), ("hpath", self.hpath), ("ref_traj", self.ref_traj), ("ref_len", ref_len),
                        ("idx_ref", self.idx_ref), ("goal", self.goal), ("polys", self.polys), ("alive", alive),
                        ("collision", collision), ("complete", complete), ("steps", steps), ("clr_dyn", clr_dyn),
                        ("clr_stc", clr_stc), ("dev_sum", dev_sum), ("dev_max", dev_max), ("n_traj", n_traj), ("traj", traj),
                        ("acts", acts), ("state_c", state_c), ("last_u_c", last_u_c), ("refs_c", refs_c), ("speed_c", speed_c),
                        ("dyn_c", dyn_c), ("U", self.U), ("y", self.y)):
            assert t.is_contiguous(), name
            setattr(a, name, t.data_ptr())
        solve_ms = []
        n_steps_run = 0
        for kt in range(max_steps):
            if not bool(alive.any()):
                break
            n_steps_run = kt + 1
            idx = torch.nonzero(alive, as_tuple=False).squeeze(1).contiguous() if self.compact else None
            nA = int(idx.numel()) if idx is not None else B
            full = nA == B
            Pa = self.P if full else self.P[:nA]
            Ua = self.U if full else self._Ua[:nA]
            ya = self.y if full else ya_buf[:nA]
            st = None
            if self.stagger_replay is not None:
                st = self.stagger_replay.pop(0).to(tdt).contiguous()
            elif self.stagger > 0:
                sign = torch.randint(0, 2, (B, H), generator=self.gen, device=dev) * 2 - 1
                mag = torch.randint(0, 11, (B, H), generator=self.gen, device=dev).to(tdt) / 10
                st = (sign.to(tdt) * mag * self.stagger).contiguous()
            a.n_run, a.step = nA, kt
            a.run = None if full else idx.data_ptr()
            a.stagger = None if st is None else st.data_ptr()
            a.U_c, a.y_c, a.gather_y = Ua.data_ptr(), ya.data_ptr(), int(not full)
            self.h.loop_step(self.dt, a, post=False)
            self.h.assemble_params(self.dt, nA, Pa, last_u_c, state_c, refs_c, speed_c, self.tuning, self.stcw, self.dynw,
                                   self.polys, dyn_c[:nA])
            if self.on_params is not None:
                self.on_params(kt, idx, Pa)
            if record is not None:
                rec = dict(robot=self.robot.cpu().numpy(), humans=self.humans.cpu().numpy(), alive=alive.bool().cpu().numpy(),
                           y_in=self.y.cpu().numpy())
            u0 = None
            if self.warm_start and kt > 0:
                shifted = torch.cat([self.U[:, 2:], self.U[:, -2:]], dim=1)
                u0 = shifted.contiguous() if full else shifted.index_select(0, idx).contiguous()
            hs = self.h
            if self._h_by_waves:      # latency-kernel family: W by the size of the running batch (results unchanged)
                cap = 3 * self._n_simd if self.dt == np.float32 else 2 * self._n_simd     # resident wavefronts
                hs = self._h_by_waves[min(4, max(2, cap // max(nA, 1)))]
            if self.dispatch_by_history and nA >= self.dispatch_min_batch:
                if kt > 0:
                    prev = self._evals if full else self._evals.index_select(0, idx)
                    hs.set_dispatch_order(torch.argsort(prev, descending=True, stable=True).to(torch.int32))
                info = self._info[:nA]
            else:
                info = None
            if self.count_status:
                st_buf = self.status[:nA]
                st_buf.fill_(-2)      # (a solver put into _solve that does not report statuses leaves them uncounted)
                self._solve(hs, Pa, nA, Ua, u0, ya, kt > 0, info, status=st_buf)
                self.status_counts.append(torch.bincount(st_buf.clamp(min=-1, max=3) + 1, minlength=5)[1:])
            else:
                self._solve(hs, Pa, nA, Ua, u0, ya, kt > 0, info)
            if info is not None:
                if full:
                    self._evals.copy_(info[:, 4])
                else:
                    self._evals.index_copy_(0, idx, info[:, 4].contiguous())
            self.h.loop_step(self.dt, a, post=True)
            if record is not None:
                Pfull = self.P if full else torch.zeros_like(self.P).index_copy_(0, idx, Pa)
                rec.update(P=Pfull.cpu().numpy(), U=self.U.cpu().numpy())
                record.append(rec)
            if self.time_solves:
                solve_ms.append(hs.last_kernel_ms())
        alive_b, collision_b, complete_b = alive.bool(), collision.bool(), complete.bool()
        collision_b = collision_b | alive_b                                      # time-out, main_base.py:407-410
        T_run = n_steps_run
        A = acts[:, :T_run]
        # rows after a scenario's end repeat its last state
        trj = traj[:, :T_run + 1]
        last = steps.clamp(max=T_run)
        t_idx = torch.minimum(torch.arange(T_run + 1, device=dev)[None, :], last[:, None])
        trj = torch.gather(trj, 1, t_idx[..., None].expand(-1, -1, 3))
        smooth = action_smoothness(A)
        return EvaluationResult(
            collision=collision_b.cpu().numpy(), complete=complete_b.cpu().numpy(), steps=steps.cpu().numpy(),
            smoothness=smooth.cpu().numpy(), clearance=clr_stc.cpu().numpy(), clearance_dyn=clr_dyn.cpu().numpy(),
            deviation=torch.stack([dev_sum / n_traj, dev_max], dim=1).cpu().numpy(),
            trajectory=trj.cpu().numpy(), actions=A.cpu().numpy(), solve_ms=solve_ms)

    def _run_torch(self, max_steps: int = 120, record: Optional[list] = None) -> EvaluationResult:
        """The same time step written out as torch expressions (rounds 1-2): the independent check of the kernels."""
        torch = self.torch
        B, N, ts = self.B, self.N, self.ts
        alive = torch.ones(B, dtype=torch.bool, device=self.dev)
        collision = torch.zeros(B, dtype=torch.bool, device=self.dev)
        complete = torch.zeros(B, dtype=torch.bool, device=self.dev)
        steps = torch.zeros(B, dtype=torch.long, device=self.dev)
        last_u = torch.zeros(B, 2, dtype=self.tdt, device=self.dev)
        traj = [self.robot.clone()]
        acts, solve_ms = [], []
        clr_dyn = torch.full((B,), float("inf"), dtype=self.tdt, device=self.dev)
        clr_stc = self._polygon_distance(self.robot[:, :2])
        d0 = torch.cdist(self.robot[:, None, :2], self.ref_traj[:, :, :2])[:, 0]
        dev_sum = d0.min(dim=1).values.clone()      # trajectory metrics include the start state (robot.past_traj[0])
        dev_max = dev_sum.clone()
        n_traj = torch.ones(B, dtype=self.tdt, device=self.dev)
        for kt in range(max_steps):
            if not bool(alive.any()):
                break
            dyn = self._predict_cv()
            refs = self._ref_states()
            dist_goal = torch.hypot(self.robot[:, 0] - self.goal[:, 0], self.robot[:, 1] - self.goal[:, 1])
            near = dist_goal < self.base_speed * N * ts
            speed = torch.where(near, torch.clamp(dist_goal / N / ts, min=self.lin_vel_max),
                                torch.full_like(dist_goal, self.base_speed))     # sic: max(), trajectory_tracker.py:308-309
            # large batches: only the scenarios still running are assembled and solved (the finished ones keep their
            # last row of U). Small batches end with their slowest instance anyway, compaction would only add host work.
            idx = torch.nonzero(alive, as_tuple=False).squeeze(1) if self.compact else None
            nA = int(idx.numel()) if idx is not None else B
            full = nA == B
            sel = (lambda x: x.contiguous()) if full else (lambda x: x.index_select(0, idx).contiguous())
            Pa = self.P if full else self.P[:nA]
            Ua = self.U if full else self._Ua[:nA]
            ya = self.y if full else self.y.index_select(0, idx).contiguous()
            self.h.assemble_params(self.dt, nA, Pa, sel(last_u), sel(self.robot), sel(refs), sel(speed),
                                   self.tuning, self.stcw, self.dynw, self.polys, sel(dyn))
            if self.on_params is not None:
                self.on_params(kt, idx, Pa)
            if record is not None:
                rec = dict(robot=self.robot.cpu().numpy(), humans=self.humans.cpu().numpy(), alive=alive.cpu().numpy(),
                           y_in=self.y.cpu().numpy())
            u0 = None
            if self.warm_start and kt > 0:
                u0 = sel(torch.cat([self.U[:, 2:], self.U[:, -2:]], dim=1))
            # longest first: the evaluation counts of the previous time step rank this step's solves (a scenario that
            # was hard a moment ago still is); pure scheduling -- the results do not depend on the order
            hs = self.h
            if self._h_by_waves:      # latency-kernel family: W by the size of the running batch (results unchanged)
                cap = 3 * self._n_simd if self.dt == np.float32 else 2 * self._n_simd     # resident wavefronts
                hs = self._h_by_waves[min(4, max(2, cap // max(nA, 1)))]
            if self.dispatch_by_history and nA >= self.dispatch_min_batch:
                if kt > 0:
                    prev = self._evals if full else self._evals.index_select(0, idx)
                    hs.set_dispatch_order(torch.argsort(prev, descending=True, stable=True).to(torch.int32))
                info = self._info[:nA]
            else:
                info = None
            if self.count_status:
                st_buf = self.status[:nA]
                st_buf.fill_(-2)      # (a solver put into _solve that does not report statuses leaves them uncounted)
                self._solve(hs, Pa, nA, Ua, u0, ya, kt > 0, info, status=st_buf)
                self.status_counts.append(torch.bincount(st_buf.clamp(min=-1, max=3) + 1, minlength=5)[1:])
            else:
                self._solve(hs, Pa, nA, Ua, u0, ya, kt > 0, info)
            if info is not None:
                if full:
                    self._evals.copy_(info[:, 4])
                else:
                    self._evals.index_copy_(0, idx, info[:, 4].contiguous())
            if not full:
                self.U.index_copy_(0, idx, Ua)
                self.y.index_copy_(0, idx, ya)
            if record is not None:
                Pfull = self.P if full else torch.zeros_like(self.P).index_copy_(0, idx, Pa)
                rec.update(P=Pfull.cpu().numpy(), U=self.U.cpu().numpy())
                record.append(rec)
            solve_ms.append(hs.last_kernel_ms())
            raw = self.U[:, :2].clone()
            act = torch.where((raw[:, 0:1] < 0), torch.zeros_like(raw), raw)     # no-backward, main_base.py:320-321
            new_robot = unicycle_rk4_step(self.robot, act, ts)
            self.robot = torch.where(alive[:, None], new_robot, self.robot)
            last_u = torch.where(alive[:, None], raw, last_u)
            self._step_humans()
            steps = steps + alive.long()
            traj.append(self.robot.clone())
            acts.append(torch.where(alive[:, None], raw, torch.full_like(raw, float("nan"))))
            # metrics and flags (main_base.py:326-335)
            dd = min_dynamic_distance(self.robot[:, :2], self.humans)
            clr_dyn = torch.where(alive, torch.minimum(clr_dyn, dd), clr_dyn)
            clr_stc = torch.where(alive, torch.minimum(clr_stc, self._polygon_distance(self.robot[:, :2])), clr_stc)
            dref = deviation_to_reference(self.robot[:, :2], self.ref_traj, self.ref_len)
            dev_sum = dev_sum + torch.where(alive, dref, torch.zeros_like(dref))
            dev_max = torch.where(alive, torch.maximum(dev_max, dref), dev_max)
            n_traj = n_traj + alive.to(self.tdt)
            col = alive & (self._in_polygon(self.robot[:, :2]) | (dd <= HUMAN_SIZE))
            done = alive & ~col & ((self.robot[:, 0] - self.goal[:, 0]).abs() <= 0.5) & \
                ((self.robot[:, 1] - self.goal[:, 1]).abs() <= 0.5) & (act[:, 0].abs() < 0.4)
            collision |= col
            complete |= done
            alive = alive & ~col & ~done
        collision |= alive                                                       # time-out, main_base.py:407-410
        A = torch.stack(acts, dim=1) if acts else torch.zeros(B, 0, 2, dtype=self.tdt, device=self.dev)
        smooth = action_smoothness(A)
        return EvaluationResult(
            collision=collision.cpu().numpy(), complete=complete.cpu().numpy(), steps=steps.cpu().numpy(),
            smoothness=smooth.cpu().numpy(), clearance=clr_stc.cpu().numpy(), clearance_dyn=clr_dyn.cpu().numpy(),
            deviation=torch.stack([dev_sum / n_traj, dev_max], dim=1).cpu().numpy(),
            trajectory=torch.stack(traj, dim=1).cpu().numpy(), actions=A.cpu().numpy(), solve_ms=solve_ms)

    def close(self):
        for h in set(self._h_by_waves.values()) | {self.h}:
            h.close()
