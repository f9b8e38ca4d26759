"""``TrajectoryTracker``: the stateful harness around one solver call per control step.

Mirror of the reference class ``/root/reference/src/pkg_mpc_tracker/trajectory_tracker.py:18-416`` -- same public
methods, argument meaning, return values and quirks -- so that ``interfaces/mpc_interface.py:20-102`` and
``main_base.py:308-311`` can drive it unchanged; the solver behind it is the HIP library instead of the OpEn
``.so``. Behaviour is pinned by ``tests/golden/tracker_harness.json`` (recorded from the reference class itself).

Quirks kept on purpose:
* ``set_work_mode(mode)`` runs on every ``run_step``; 'work' => base speed 0.8 * lin_vel_max;
* close to the goal the reference speed is ``max(dist/N/ts, lin_vel_max)`` (reference :305-310 uses ``max``);
* the state after the step is always propagated from the state at entry (only exact for action_steps = 1);
* ``pred_states`` re-applies u_0..u_{N-1} starting from the propagated state.
"""
from __future__ import annotations

import math
import os
import sys
from typing import Callable, List, Optional, Tuple, Union

import numpy as np


class TrajectoryTracker:
    def __init__(self, config, robot_specification, use_tcp: bool = False, verbose: bool = False,
                 solver_factory: Optional[Callable] = None):
        self._prt_name = "[TrajTracker]"
        self.vb = verbose
        self.config = config
        self.robot_spec = robot_specification
        self.ts, self.ns, self.nu, self.N_hor = config.ts, config.ns, config.nu, config.N_hor
        self.idle = True
        self.set_work_mode(mode="safe")
        self.set_obstacle_weights(stc_weights=10, dyn_weights=10)
        self.use_tcp = use_tcp
        if use_tcp:     # reference :62-66: OptimizerTcpManager(solver_path).start(); ping()
            from .tcp import OptimizerTcpManager
            path = os.path.join("", self.config.build_directory, self.config.optimizer_name)
            self.mng = OptimizerTcpManager(path, solver_factory=lambda: self._load_solver(solver_factory))
            self.mng.start()
            self.mng.ping()
        else:
            self.solver = self._load_solver(solver_factory)

    # ------------------------------------------------------------------------------------------------------
    def _load_solver(self, solver_factory):
        """Reference :54-66 imports ``<build_directory>/<optimizer_name>/<optimizer_name>`` relative to the CWD and
        calls its ``solver()``. ``solver_build.build()`` writes exactly such a module; if it has not been run the
        solver is created directly from the configuration."""
        if solver_factory is not None:
            return solver_factory()
        name = self.config.optimizer_name
        path = os.path.join("", self.config.build_directory, name)
        if os.path.isfile(os.path.join(path, name + ".py")):
            if path not in sys.path:
                sys.path.append(path)
            return __import__(name).solver()
        from .solver import make_config, solver
        return solver(make_config(self.config, self.robot_spec))

    def load_motion_model(self, motion_model: Callable) -> None:
        """``s' = f(s, a, ts)``"""
        self.motion_model = motion_model

    def load_init_states(self, current_state: np.ndarray, goal_state: np.ndarray):
        if not isinstance(current_state, np.ndarray) or not isinstance(goal_state, np.ndarray):
            raise TypeError(f"State should be numpy.ndarry, got {type(current_state)}/{type(goal_state)}.")
        self.state = current_state
        self.final_goal = goal_state
        self.past_states: List[np.ndarray] = []
        self.past_actions: List[np.ndarray] = []
        self.cost_timelist: List[float] = []
        self.solver_time_timelist: List[float] = []
        self.idx_ref_traj = 0
        self.idx_ref_path = 0
        self.idle = False

    def set_obstacle_weights(self, stc_weights: Union[list, int], dyn_weights: Union[list, int]):
        def expand(w):
            if isinstance(w, list):
                return w
            if isinstance(w, (float, int)):
                return [w] * self.N_hor
            raise TypeError(f"Unsupported datatype for obstacle weights, got {type(w)}.")
        self.stc_weights = expand(stc_weights)
        self.dyn_weights = expand(dyn_weights)

    def set_work_mode(self, mode: str = "safe"):
        """'aligning' (half speed, heading weight only), 'safe' (20 %), 'work' (80 %), 'super' (100 %)."""
        vmax = self.robot_spec.lin_vel_max
        if mode == "aligning":
            self.base_speed = vmax * 0.5
            self.tuning_params = [0.0] * self.config.nq
            self.tuning_params[2] = 100
            return
        c = self.config
        self.tuning_params = [c.qpos, c.qvel, c.qtheta, c.lin_vel_penalty, c.ang_vel_penalty,
                              c.qpN, c.qthetaN, c.qrpd, c.lin_acc_penalty, c.ang_acc_penalty]
        scale = {"safe": 0.2, "work": 0.8, "super": 1.0}
        if mode not in scale:
            raise ModuleNotFoundError(f"There is no mode called {mode}.")
        self.base_speed = vmax * scale[mode]

    def set_current_state(self, current_state: np.ndarray):
        if not isinstance(current_state, np.ndarray):
            raise TypeError(f"State should be numpy.ndarry, got {type(current_state)}.")
        self.state = current_state

    def set_ref_trajectory(self, ref_path: List[tuple], ref_traj: List[tuple] = None):
        self.idx_ref_path = 0
        self.idx_ref_traj = 0
        self.ref_path = ref_path
        self.ref_traj = ref_traj if ref_traj is not None else \
            self.get_ref_traj(self.ts, ref_path, self.state, self.base_speed)

    def set_ref_states(self, ref_states: np.ndarray = None) -> np.ndarray:
        if ref_states is not None:
            self.ref_states = ref_states
        else:
            self.ref_states, self.idx_ref_traj = self.get_ref_states(self.idx_ref_traj, self.ref_traj, self.state,
                                                                     self.N_hor)
        return self.ref_states

    def check_termination_condition(self, state: np.ndarray, action: np.ndarray, final_goal: np.ndarray) -> bool:
        done = bool(np.allclose(state[:2], final_goal[:2], atol=0.5, rtol=0) and abs(action[0]) < 0.4)
        if done:
            self.idle = True
            if self.vb:
                print(f"{self._prt_name} MPC solution found.")
        return done

    # ------------------------------------------------------------------------------------------------------
    @staticmethod
    def get_ref_traj(ts: float, ref_path: List[tuple], state: tuple, speed: float) -> List[tuple]:
        """Constant-speed resampling of the way-point path: one (x, y, heading) per ``ts`` (reference :202-240).

        A point is emitted after every full ``ts`` of travel; reaching a way-point inside a step switches the
        target without emitting, and the walk ends when the last way-point is reached.
        """
        x, y = state[0], state[1]
        idx = 0
        tx, ty = ref_path[0][0], ref_path[0][1]
        traj: List[tuple] = []
        step = speed * ts
        while True:
            emit = False
            while True:
                dist = math.hypot(tx - x, ty - y)
                if dist < 1e-9:                       # standing on the way-point: take the next one
                    idx += 1
                    tx, ty = ref_path[idx][0], ref_path[idx][1]
                    break
                dx, dy = (tx - x) / dist, (ty - y) / dist
                if dist / speed > ts:                 # a full step towards the way-point
                    x, y = x + dx * step, y + dy * step
                    emit = True
                    break
                x, y = x + dx * speed * (dist / speed), y + dy * speed * (dist / speed)
                idx += 1
                if idx > len(ref_path) - 1:
                    return traj + [(x, y, math.atan2(dy, dx))]
                tx, ty = ref_path[idx][0], ref_path[idx][1]
            if emit:
                traj.append((x, y, math.atan2(dy, dx)))

    @staticmethod
    def get_ref_states(idx_ref_traj: int, ref_traj: List[tuple], state: tuple, action_steps=1, horizon=20
                       ) -> Tuple[np.ndarray, int]:
        """Closest trajectory point within a sliding window (-1 ... +5 action steps), then ``horizon`` rows padded with
        the last one (reference :242-270). NOTE the reference calls this with ``N_hor`` in the ``action_steps``
        position (:186-187), so the window is [-N_hor, +5 N_hor) and ``horizon`` keeps its default 20."""
        arr = np.array(ref_traj)
        lo = max(0, idx_ref_traj - 1 * action_steps)
        hi = min(len(ref_traj), idx_ref_traj + 5 * action_steps)
        d = [math.hypot(state[0] - q[0], state[1] - q[1]) for q in ref_traj[lo:hi]]
        idx_next = d.index(min(d)) + lo
        rows = arr[idx_next:idx_next + horizon]
        if idx_next + horizon >= len(arr):
            pad = horizon - (len(arr) - idx_next)
            rows = np.concatenate([arr[idx_next:], np.repeat(arr[-1:], pad, axis=0)], axis=0)
        return np.array(rows[:, :3], dtype=float), idx_next

    # ------------------------------------------------------------------------------------------------------
    def run_step(self, stc_constraints: list, dyn_constraints: list, other_robot_states: list = None,
                 ref_states: np.ndarray = None, mode: str = "safe"):
        """One control step: returns ``(actions, pred_states, ref_states, cost)`` or ``-1`` on a solver error."""
        self.set_work_mode(mode)
        cfg, N = self.config, self.N_hor
        if stc_constraints is None:
            stc_constraints = [0] * (cfg.Nstcobs * cfg.nstcobs)
        if dyn_constraints is None:
            dyn_constraints = [0] * (cfg.Ndynobs * cfg.ndynobs * (N + 1))
        if other_robot_states is None:
            other_robot_states = [0] * (self.ns * (N + 1) * cfg.Nother)

        ref_states = self.set_ref_states(ref_states)
        goal_row = ref_states[-1, :]

        dist_to_goal = math.hypot(self.state[0] - self.final_goal[0], self.state[1] - self.final_goal[1])
        if dist_to_goal >= self.base_speed * N * self.ts:
            speed_ref = self.base_speed
        else:
            speed_ref = max(dist_to_goal / N / self.ts, self.robot_spec.lin_vel_max)   # sic: max, as the reference

        last_u = self.past_actions[-1] if len(self.past_actions) else np.zeros(self.nu)
        params = list(last_u) + list(self.state) + list(goal_row) + self.tuning_params + \
            ref_states.reshape(-1).tolist() + [speed_ref] * N + other_robot_states + \
            stc_constraints + dyn_constraints + self.stc_weights + self.dyn_weights
        try:
            taken, pred_states, actions, cost, solver_time, exit_status = \
                self.run_solver(params, self.state, cfg.action_steps)
        except RuntimeError as err:
            print(f"Fatal: Cannot run solver. {err}.")
            return -1

        self.past_states.append(self.state)
        self.past_states += taken[:-1]
        self.past_actions += actions
        self.state = taken[-1]
        self.cost_timelist.append(cost)
        self.solver_time_timelist.append(solver_time)
        if exit_status in cfg.bad_exit_codes and self.vb:
            print(f"{self._prt_name} Bad converge status: {exit_status}")
        return actions, pred_states, ref_states, cost

    def run_solver(self, parameters: list, state: np.ndarray, take_steps: int = 1):
        """Solve, then roll the returned controls out with the motion model (reference :339-383; over the socket
        :385-400)."""
        if self.use_tcp:
            resp = self.mng.call(parameters)
            if not resp.is_ok():
                err = resp.get()
                self.mng.kill()
                raise RuntimeError(f"MPC Solver error: [{err.code}]{err.message}")
            sol = resp.get()
        else:
            sol = self.solver.run(parameters)
            if sol is None:
                raise RuntimeError("MPC Solver error: the solver returned no solution")
        u, nu = sol.solution, self.nu
        taken = [self.motion_model(state, np.array(u[i * nu:(i + 1) * nu]), self.ts) for i in range(take_steps)]
        pred = [taken[-1]]
        for i in range(len(u) // nu):
            pred.append(self.motion_model(pred[-1], np.array(u[i * nu:i * nu + 2]), self.ts))
        actions = [np.array(a) for a in np.array(u[:nu * take_steps]).reshape(take_steps, nu).tolist()]
        return taken, pred[1:], actions, sol.cost, sol.solve_time_ms, sol.exit_status
