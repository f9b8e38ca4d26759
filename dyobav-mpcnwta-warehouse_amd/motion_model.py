"""Unicycle motion model used by the tracker harness after the solve (reference:
``src/basic_motion_model/motion_model.py:141-163`` and ``UnicycleModel`` :85-99), numpy only.

RK4 of s' = (v cos th, v sin th, w) with a constant action only ever evaluates the heading at th, th+h, th+h and
th+2h (h = ts*w/2), so one step is
    th+ = th + ts*w ;  x+ = x + ts*v*(cos th + 4 cos(th+h) + cos(th+2h))/6 ;  y+ likewise with sin
-- the same closed form the HIP kernel uses for the rollout.
"""
from __future__ import annotations

import numpy as np


def unicycle_model(state: np.ndarray, action: np.ndarray, ts: float, rk4: bool = True) -> np.ndarray:
    x, y, th = (float(v) for v in state[:3])
    v, w = float(action[0]), float(action[1])
    if not rk4:
        return np.array([x + ts * v * np.cos(th), y + ts * v * np.sin(th), th + ts * w])
    h = 0.5 * ts * w
    c = (np.cos(th) + 4.0 * np.cos(th + h) + np.cos(th + 2.0 * h)) / 6.0
    s = (np.sin(th) + 4.0 * np.sin(th + h) + np.sin(th + 2.0 * h)) / 6.0
    return np.array([x + ts * v * c, y + ts * v * s, th + ts * w])


class UnicycleModel:
    """Callable ``s_next = model(state, action)`` with the sampling time fixed at construction."""

    def __init__(self, sampling_time: float, rk4: bool = True) -> None:
        self.ts = sampling_time
        self.rk4 = rk4
        self.state_dim, self.action_dim = 3, 2

    def __call__(self, state: np.ndarray, action: np.ndarray, ts: float = None) -> np.ndarray:
        if ts is not None:
            self.ts = ts
        return unicycle_model(state, action, self.ts, rk4=self.rk4)
