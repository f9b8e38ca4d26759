"""yaml -> typed configuration objects, with the attribute names of the reference's ``src/configs.py``
(``Configurator`` :10-42, ``CircularRobotSpecification`` :86-103, ``MpcConfiguration`` :140-176) so that code
written against the reference (``TrajectoryTracker``, ``MpcInterface``) can consume them unchanged.
Only the two classes the solve path reads are provided.
"""
from __future__ import annotations

import yaml

ROBOT_KEYS = ("ts", "vehicle_width", "vehicle_margin", "social_margin", "lin_vel_min", "lin_vel_max",
              "lin_acc_min", "lin_acc_max", "ang_vel_max", "ang_acc_max")
MPC_KEYS = ("ts", "N_hor", "action_steps", "ns", "nu", "nq", "Nother", "nstcobs", "Nstcobs", "ndynobs", "Ndynobs",
            "max_solver_time", "build_directory", "build_type", "bad_exit_codes", "optimizer_name",
            "lin_vel_penalty", "lin_acc_penalty", "ang_vel_penalty", "ang_acc_penalty", "qrpd", "qpos", "qvel",
            "qtheta", "qpN", "qthetaN")


class Configurator:
    """Every top-level yaml key becomes an attribute (reference: configs.py:10-22)."""

    def __init__(self, yaml_fp: str, with_partition: bool = False) -> None:
        with open(yaml_fp, "r") as stream:
            if with_partition:
                data = {}
                for doc in yaml.safe_load_all(stream):
                    data.update(doc or {})
            else:
                data = yaml.safe_load(stream)
        self.yaml_path = yaml_fp
        for key, value in data.items():
            setattr(self, key, value)


class _Section:
    KEYS: tuple = ()

    def __init__(self, config: Configurator) -> None:
        self._config = config
        missing = [k for k in self.KEYS if not hasattr(config, k)]
        if missing:
            raise AttributeError(f"{getattr(config, 'yaml_path', 'config')}: missing keys {missing}")
        for k in self.KEYS:
            setattr(self, k, getattr(config, k))

    @classmethod
    def from_yaml(cls, yaml_fp: str, with_partition: bool = False):
        return cls(Configurator(yaml_fp, with_partition))


class CircularRobotSpecification(_Section):
    """ts, vehicle_width/margin, social_margin and the velocity / acceleration limits."""
    KEYS = ROBOT_KEYS


class MpcConfiguration(_Section):
    """Horizon, dimensions, weights and solver build options."""
    KEYS = MPC_KEYS
