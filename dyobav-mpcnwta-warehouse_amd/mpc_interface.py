"""``MpcInterface``: the adapter between the simulator and the trajectory tracker.

Mirror of the reference class ``/root/reference/src/interfaces/mpc_interface.py:20-102`` (same constructor arguments,
``set_current_state`` / ``update_map`` / ``update_global_path`` / ``run_step`` and the same 5-tuple returned by
``run_step``), so that ``main_base.py:308-311`` can drive it unchanged. The static-obstacle marshalling
(closest ``Nstcobs`` map polygons -> half-space rows, reference :73-80, :90-100 + ``utils_geo.py``) runs on the device
through ``nmpc_assemble_params`` (f1) -- there is no host implementation of it in this package; the dynamic-obstacle
flattening (:82-88) is a list copy and stays on the host like in the reference.

Limitations vs the reference: map polygons must be convex quadrilaterals given in vertex order (what the reference's
map pipeline produces after ``cvt_occ2geo``); the order of the returned ``closest_obstacle_list`` is nearest-first
(the reference's ``argpartition`` order is unspecified).
"""
from __future__ import annotations

import itertools
import os
from typing import Callable, List, Optional, Tuple

import numpy as np

from . import _capi
from .configs import CircularRobotSpecification, MpcConfiguration
from .motion_model import UnicycleModel
from .solver import make_config
from .trajectory_tracker import TrajectoryTracker

ROOT_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class MpcInterface:
    def __init__(self, config_file_name: str, current_state: np.ndarray, geo_map, verbose: bool = True,
                 solver_factory: Optional[Callable] = None) -> None:
        self._prt_name = "MPCInterface"
        path = config_file_name if os.path.isabs(config_file_name) else os.path.join(ROOT_DIR, "config", config_file_name)
        self.config_mpc = MpcConfiguration.from_yaml(path)
        self.config_robot = CircularRobotSpecification.from_yaml(path)
        self.traj_tracker = TrajectoryTracker(self.config_mpc, self.config_robot, verbose=verbose,
                                              solver_factory=solver_factory)
        self.traj_tracker.load_motion_model(UnicycleModel(self.config_robot.ts))
        self.state = current_state
        self.geo_map = geo_map
        self.prepared = False
        self._asm = None          # device handle for the static-obstacle marshalling (created on first use)
        self._map_dev = None
        self._map_src = None

    def set_current_state(self, current_state: np.ndarray):
        self.state = current_state
        self.traj_tracker.set_current_state(current_state)

    def update_map(self, geo_map):
        self.geo_map = geo_map

    def update_global_path(self, new_global_path: List[tuple]):
        self.traj_tracker.load_init_states(self.state, np.array(new_global_path[-1]))
        self.traj_tracker.set_work_mode("work")
        self.traj_tracker.set_ref_trajectory(new_global_path)
        self.ref_path = new_global_path
        self.ref_traj = self.traj_tracker.ref_traj
        self.base_speed = self.traj_tracker.base_speed
        self.prepared = True

    def run_step(self, mode, full_dyn_obstacle_list: list = None, map_updated: bool = True
                 ) -> Tuple[List[np.ndarray], List[np.ndarray], float, List[List[tuple]], np.ndarray]:
        """Returns ``(actions, pred_states, cost, closest_obstacle_list, current_refs)``."""
        if not self.prepared:
            raise ValueError("MPCInterface is not prepared. Call update_global_path() first.")
        if map_updated or not hasattr(self, "_stc_cache"):
            self._stc_cache = self.get_stc_constraints()
        stc_constraints, closest_obstacle_list = self._stc_cache
        dyn_constraints = self.get_dyn_constraints(full_dyn_obstacle_list)
        actions, self.pred_states, current_refs, cost = self.traj_tracker.run_step(stc_constraints, dyn_constraints,
                                                                                   mode=mode)
        self.state = self.traj_tracker.state
        return actions, self.pred_states, cost, closest_obstacle_list, current_refs

    # ------------------------------------------------------------------------------------------------------
    def get_stc_constraints(self) -> Tuple[list, List[List[tuple]]]:
        """(b, a0, a1) rows of the ``Nstcobs`` map polygons closest to the robot, computed by the f1 kernels."""
        import torch
        cfg = self.config_mpc
        polys = self.geo_map.processed_obstacle_list
        if self._asm is None:
            self._asm = _capi.Handle(make_config(cfg, self.config_robot))
        if self._map_src is not polys:      # upload the static map once per map object
            arr = np.asarray(polys, dtype=np.float64)
            if arr.ndim != 3 or arr.shape[1:] != (4, 2):
                raise ValueError("map polygons must be quadrilaterals: shape [M, 4, 2]")
            self._map_dev, self._map_src = torch.from_numpy(np.ascontiguousarray(arr)).cuda(), polys
        N, dt = cfg.N_hor, torch.float64
        z = lambda *s: torch.zeros(*s, dtype=dt, device="cuda")
        P = torch.empty(1, self._asm.np_, dtype=dt, device="cuda")
        sel = torch.empty(1, cfg.Nstcobs, dtype=torch.int32, device="cuda")
        state = torch.from_numpy(np.asarray(self.state, dtype=np.float64).reshape(1, 3)).cuda()
        self._asm.assemble_params(np.float64, 1, P, z(1, 2), state, z(1, N, 3), z(1), z(10), z(N), z(N),
                                  self._map_dev, None, None, sel)
        torch.cuda.synchronize()
        off = 18 + 4 * N + 3 * cfg.Nother * (N + 1)
        stc = P[0, off:off + cfg.Nstcobs * cfg.nstcobs].cpu().numpy().tolist()
        closest = [polys[i] for i in sel[0].cpu().numpy().tolist() if i >= 0]
        return stc, closest

    def get_dyn_constraints(self, full_dyn_obstacle_list=None):
        per = (self.config_mpc.N_hor + 1) * self.config_mpc.ndynobs
        out = [0.0] * self.config_mpc.Ndynobs * per
        if full_dyn_obstacle_list is not None:
            for i, obstacle in enumerate(full_dyn_obstacle_list):
                out[i * per:(i + 1) * per] = list(itertools.chain(*obstacle))
        return out

    def get_closest_n_stc_obstacles(self) -> List[List[tuple]]:
        return self.get_stc_constraints()[1]
