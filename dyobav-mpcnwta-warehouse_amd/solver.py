"""The solver object the reference imports from its generated module, backed by the HIP kernels.

Reference interface being mirrored (``/root/reference/src/pkg_mpc_tracker/trajectory_tracker.py``):

* ``built_solver = __import__(optimizer_name); self.solver = built_solver.solver()``           (:58-61)
* ``solution = self.solver.run(parameters)`` with a flat Python list of floats               (:362)
* attributes read from the result: ``solution``, ``cost``, ``exit_status``, ``solve_time_ms``  (:364-367)
* stub signature ``run(p, initial_guess, initial_lagrange_multipliers, initial_penalty)``      (:13-15)

plus the batched front end (``BatchSolver``) that the reference does not have: B independent problems per
launch, inputs/outputs as numpy arrays or device tensors.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import _capi
from ._capi import EXIT_STATUS_NAMES, Handle, NmpcConfigStruct, default_config_struct

_ROBOT_FIELDS = ("ts", "lin_vel_min", "lin_vel_max", "ang_vel_max", "lin_acc_min", "lin_acc_max", "ang_acc_max",
                 "vehicle_width", "vehicle_margin", "social_margin")


# ---- yaml `max_solver_time` as an evaluation budget ------------------------------------------------------------------
# OpEn stops a solve after `max_solver_time` micro-seconds of ITS OWN wall clock (`with_max_duration_micros`,
# mpc_builder.py:189; 0.1 s in mpc_fast.yaml, 0.5 s in mpc_default.yaml) and the tracker goes on with the truncated
# answer (trajectory_tracker.py:318-335 only prints). What that cap means for the RESULT is "as many psi / grad-psi
# evaluations as one host core of the reference's machine gets through in that time" -- the GPU's own clock has nothing to
# do with it (one problem takes 1-20 ms here; a batch of 65 536 takes a second) and a wall-clock cap makes results depend
# on the batch, the clock and the co-residents. `nmpc_config.max_evaluations` is the same cap as a count; this is the
# mapping:
#     max_evaluations = max_solver_time [s] x CPU_FORWARD_FLOPS_PER_S / forward_flops(N, Nother, Nstc, Ndyn)
# with forward_flops = SURVEY.md 8(d)'s F_fwd (the work of one evaluation grows with the dimensions the yaml sets) and
# CPU_FORWARD_FLOPS_PER_S the measured rate of the generated-code-equivalent CPU path -- the fp64 oracle with cos / sin of
# every ellipse on every evaluation and cost / gradient as separate calls -- on ONE core, in evaluated points x F_fwd per
# second: tools/measure_cpu_eval_rate.py (profiles/r06_cpu_eval_rate.json: 1.44-2.04e9, median 1.64e9, over two dimension
# sets x two scenario families on this round's build container -- the flop scaling between 15 and 40 obstacle rows holds to
# ~10 %; the GPU box's host in round 5: 3.5 solves/s per core x 6 665 evaluations = 1.48e9; bench.py re-measures it on
# every run, `cpu_baseline.evals_per_s_per_core`). Shipped yaml (N = 20, 15 obstacle rows): 0.1 s -> 4 947 evaluations,
# 0.5 s -> 24 737; configs[2] (40 rows): 0.1 s -> 2 526 -- against ~6 650 evaluations for an instance that runs to its
# iteration caps and 10-300 for one that converges.
CPU_FORWARD_FLOPS_PER_S = 1.6e9


def forward_flops(N: int, Nother: int, Nstc: int, Ndyn: int) -> int:
    """Algorithmic flops of one forward evaluation of psi, SURVEY.md 8(d):
    N (33 + 8 (2 Nother - 1) + 28 Nstc + 62 Ndyn) + 10 N (N + 1) + 12 N."""
    return N * (33 + 8 * (2 * Nother - 1) + 28 * Nstc + 62 * Ndyn) + 10 * N * (N + 1) + 12 * N


def evaluation_budget(max_solver_time_us: float, N: int, Nother: int, Nstc: int, Ndyn: int,
                      cpu_forward_flops_per_s: float = CPU_FORWARD_FLOPS_PER_S) -> int:
    """`max_solver_time` (micro-seconds of the reference's CPU solver) -> `nmpc_config.max_evaluations` (see above)."""
    if not max_solver_time_us or max_solver_time_us <= 0:
        return 0
    return max(1, int(round(max_solver_time_us * 1e-6 * cpu_forward_flops_per_s / forward_flops(N, Nother, Nstc, Ndyn))))


def make_config(mpc_config=None, robot_spec=None, device_id: int = 0, time_cap: str = "evaluations",
                **overrides) -> NmpcConfigStruct:
    """``nmpc_config`` from the reference-style configuration objects (``configs.MpcConfiguration`` /
    ``configs.CircularRobotSpecification``); unspecified values are the OpEn defaults the reference builds with
    (``solver_build/mpc_builder.py:187-195``).

    ``max_solver_time`` (micro-seconds; ``with_max_duration_micros``, ``mpc_builder.py:189``), ``time_cap``:

    * ``"evaluations"`` (default) -- ``nmpc_config.max_evaluations = evaluation_budget(max_solver_time, dims)``: the
      deterministic, batch-safe form. An instance that uses its budget up stops where OpEn stops when its clock runs out
      (after the inner iteration in progress; no further outer iteration) and reports ``NotConvergedOutOfTime``
      (``bad_exit_codes`` of the yaml files, ``trajectory_tracker.py:334-335``) with the point it has reached.
    * ``"wall_clock"`` -- ``nmpc_config.max_solver_time_us``: the GPU's 100 MHz real-time counter per instance. Only
      meaningful for B = 1 (results depend on the clock and on what else runs); kept for callers that want a latency bound.
    * ``"none"`` -- iteration caps only (what the batch API's ``default_config_struct`` gives).
    """
    if time_cap not in ("evaluations", "wall_clock", "none"):
        raise ValueError(f"time_cap = {time_cap!r} (evaluations, wall_clock or none)")
    cfg = default_config_struct()
    cfg.device_id = device_id
    if mpc_config is not None:
        cfg.N_hor, cfg.Nother = int(mpc_config.N_hor), int(mpc_config.Nother)
        cfg.Nstcobs, cfg.Ndynobs = int(mpc_config.Nstcobs), int(mpc_config.Ndynobs)
        if int(mpc_config.ns) != 3 or int(mpc_config.nu) != 2 or int(mpc_config.nq) != 10 or \
                int(mpc_config.nstcobs) != 12 or int(mpc_config.ndynobs) != 6:
            raise ValueError("the kernels implement ns=3, nu=2, nq=10, nstcobs=12, ndynobs=6 (the shipped yaml values)")
        cfg.ts = float(mpc_config.ts)
        t_us = float(getattr(mpc_config, "max_solver_time", None) or 0.0)
        if t_us > 0 and time_cap == "wall_clock":
            cfg.max_solver_time_us = t_us
        elif t_us > 0 and time_cap == "evaluations":
            cfg.max_evaluations = evaluation_budget(t_us, cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs)
    if robot_spec is not None:
        for k in _ROBOT_FIELDS:
            setattr(cfg, k, float(getattr(robot_spec, k)))
    for k, v in overrides.items():
        if not hasattr(cfg, k):
            raise TypeError(f"unknown solver option {k!r}")
        setattr(cfg, k, v)
    return cfg


@dataclass
class OptimizerSolution:
    """Fields of OpEn's generated ``OptimizerSolution``."""
    exit_status: str
    num_outer_iterations: int
    num_inner_iterations: int
    last_problem_norm_fpr: float
    f1_infeasibility: float
    f2_norm: float
    solve_time_ms: float
    penalty: float
    solution: List[float]
    lagrange_multipliers: List[float]
    cost: float


class Solver:
    """One-problem-at-a-time facade: ``solver().run(p)`` == a batch of one through ``nmpc_solve_batch_*``.

    State carried between calls, as in OpEn's Python binding (recalled): when no
    ``initial_lagrange_multipliers`` are passed, the multipliers left by the previous call are reused
    (``keep_multipliers=False`` resets them to zero every call instead).
    """

    def __init__(self, config: Optional[NmpcConfigStruct] = None, dtype=np.float64, keep_multipliers: bool = True,
                 warm_start: bool = False):
        self.config = config if config is not None else default_config_struct()
        self.dtype = np.dtype(dtype)
        self.keep_multipliers = keep_multipliers
        self.warm_start = warm_start      # row f4: shifted previous solution as the initial guess when none is passed
        self._u_prev = None
        self._handle = Handle(self.config)
        self.num_parameters = self._handle.np_
        self.num_decision_variables = self._handle.n
        n = self._handle.n
        self._y = np.zeros((1, n), dtype=self.dtype)
        # persistent host buffers of the B = 1 path + "all arguments are host pointers": no per-call allocation and no
        # hipPointerGetAttributes lookups (12 per call before; the host side of a solve was 1.7 ms of 2.4 ms)
        self._handle.set_pointer_mode(1)
        self._bufs = dict(P=np.empty((1, self.num_parameters), dtype=self.dtype), U=np.empty((1, n), dtype=self.dtype),
                          cost=np.empty(1, dtype=self.dtype), status=np.empty(1, dtype=np.int32),
                          iters=np.empty((1, 2), dtype=np.int32), y=np.zeros((1, n), dtype=self.dtype),
                          info=np.empty((1, 8), dtype=self.dtype), u0=np.empty((1, n), dtype=self.dtype),
                          c0=np.empty(1, dtype=self.dtype))

    def run(self, p, initial_guess=None, initial_lagrange_multipliers=None, initial_penalty=None):
        n = self.num_decision_variables
        p = np.asarray(p).ravel()        # list, 1-D array or (1, np)-shaped array, as OpEn's binding accepts them
        if p.size != self.num_parameters:
            print(f"1600 -> wrong number of parameters: expected {self.num_parameters}, got {p.size}")
            return None
        b = self._bufs
        b["P"][0, :] = p
        u0 = None
        if initial_guess is not None:
            u0 = np.asarray(initial_guess, dtype=self.dtype).reshape(1, -1)
            if u0.shape[1] != n:
                print(f"1700 -> wrong length of initial guess: expected {n}, got {u0.shape[1]}")
                return None
        elif self.warm_start and self._u_prev is not None:
            u0 = shift_solution(self._u_prev)
        if initial_lagrange_multipliers is not None:
            y = np.asarray(initial_lagrange_multipliers, dtype=self.dtype).reshape(1, -1).copy()
            if y.shape[1] != n:
                print(f"1800 -> wrong dimension of Lagrange multipliers: expected {n}, got {y.shape[1]}")
                return None
        elif self.keep_multipliers:
            y = self._y.copy()
        else:
            y = np.zeros((1, n), dtype=self.dtype)
        c0 = None
        if initial_penalty is not None:
            b["c0"][0] = initial_penalty
            c0 = b["c0"]
        if u0 is not None:
            b["u0"][:] = u0
            u0 = b["u0"]
        b["y"][:] = y
        tic = time.perf_counter()
        self._handle.solve_raw(self.dtype, b["P"], 1, b["U"], b["cost"], b["status"], b["iters"], u0, b["y"], True, c0,
                               b["info"], True)
        wall_ms = (time.perf_counter() - tic) * 1e3
        out = dict(U=b["U"].copy(), cost=b["cost"], status=b["status"], iters=b["iters"], y=b["y"].copy(), info=b["info"])
        status = int(out["status"][0])
        if status == 3:  # OpEn: Err(NotFiniteComputation) -> binding returns None
            print("2000 -> Problem solution failed")
            return None
        self._y = out["y"].astype(self.dtype)
        self._u_prev = out["U"].astype(self.dtype)
        info = out["info"][0]
        return OptimizerSolution(
            exit_status=EXIT_STATUS_NAMES[status], num_outer_iterations=int(out["iters"][0, 0]),
            num_inner_iterations=int(out["iters"][0, 1]), last_problem_norm_fpr=float(info[0]),
            f1_infeasibility=float(info[2] / info[3]) if info[3] else float(info[2]), f2_norm=float(info[1]),
            solve_time_ms=wall_ms, penalty=float(info[3]), solution=[float(v) for v in out["U"][0]],
            lagrange_multipliers=[float(v) for v in out["y"][0]], cost=float(out["cost"][0]))

    def run_many(self, P) -> list:
        """All rows of ``P`` in one launch (zero initial guess and multipliers each, like independent ``run`` calls
        on fresh solver objects); one ``OptimizerSolution`` per row. Used by the ``RunBatch`` request of ``tcp``."""
        P = np.asarray(P, dtype=self.dtype).reshape(-1, self.num_parameters)
        tic = time.perf_counter()
        out = self._handle.solve(P, dtype=self.dtype)
        wall_ms = (time.perf_counter() - tic) * 1e3
        sols = []
        for b in range(P.shape[0]):
            info = out["info"][b]
            sols.append(OptimizerSolution(
                exit_status=EXIT_STATUS_NAMES[int(out["status"][b])], num_outer_iterations=int(out["iters"][b, 0]),
                num_inner_iterations=int(out["iters"][b, 1]), last_problem_norm_fpr=float(info[0]),
                f1_infeasibility=float(info[2] / info[3]) if info[3] else float(info[2]), f2_norm=float(info[1]),
                solve_time_ms=wall_ms, penalty=float(info[3]), solution=[float(v) for v in out["U"][b]],
                lagrange_multipliers=[float(v) for v in out["y"][b]], cost=float(out["cost"][b])))
        return sols

    def close(self):
        self._handle.close()


def shift_solution(U: np.ndarray) -> np.ndarray:
    """Receding-horizon shift of solutions ``[B, 2N]`` laid out (v0, w0, v1, w1, ...): drop the applied action, repeat
    the last one. Not something the reference does (``trajectory_tracker.py:362`` passes no initial guess)."""
    U = np.asarray(U)
    return np.concatenate([U[:, 2:], U[:, -2:]], axis=1)


def solver(config: Optional[NmpcConfigStruct] = None, dtype=np.float64, **kwargs) -> Solver:
    """Factory with the name the generated OpEn module exports (``<optimizer_name>.solver()``)."""
    return Solver(config, dtype=dtype, **kwargs)


@dataclass
class BatchResult:
    U: np.ndarray
    cost: np.ndarray
    status: np.ndarray
    iters: np.ndarray
    y: np.ndarray
    info: np.ndarray
    kernel_ms: float = 0.0
    exit_status: List[str] = field(default_factory=list)


class BatchSolver:
    """B independent MPC problems per launch (robots x Monte-Carlo scenarios).

    ``dispatch_by_history`` (default on, batches of >= ``dispatch_min_batch`` problems): the evaluation counts of a call
    rank the problems of the next call of the same size, longest first (``nmpc_set_dispatch_order``) -- in a
    receding-horizon loop problem i of this time step resembles problem i of the previous one, and a launch that starts
    its long solves first drains sooner. Pure scheduling: the results do not depend on it."""

    def __init__(self, config: Optional[NmpcConfigStruct] = None, dtype=np.float32, dispatch_by_history: bool = True,
                 dispatch_min_batch: int = 32768):
        self.config = config if config is not None else default_config_struct()
        self.dtype = np.dtype(dtype)
        self.handle = Handle(self.config)
        self.num_parameters = self.handle.np_
        self.num_decision_variables = self.handle.n
        self.dispatch_by_history, self.dispatch_min_batch = dispatch_by_history, dispatch_min_batch

    def run_batch(self, P: np.ndarray, u0=None, y0=None, c0=None) -> BatchResult:
        out = self.handle.solve(np.asarray(P), u0=u0, y0=y0, c0=c0, dtype=self.dtype)
        if self.dispatch_by_history and out["U"].shape[0] >= self.dispatch_min_batch:
            self.handle.set_dispatch_order(np.argsort(-out["info"][:, 4], kind="stable").astype(np.int32))
        return BatchResult(out["U"], out["cost"], out["status"], out["iters"], out["y"], out["info"],
                           self.handle.last_kernel_ms(), [EXIT_STATUS_NAMES[int(s)] for s in out["status"]])

    def close(self):
        self.handle.close()


__all__ = ["make_config", "OptimizerSolution", "Solver", "solver", "BatchSolver", "BatchResult", "_capi"]
