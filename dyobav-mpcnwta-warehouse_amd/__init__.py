"""MI355X-native batched NMPC solver: drop-in for the OpEn/PANOC solver of DyObAv-MPCnWTA-Warehouse.

Host side (Python, mirrors the reference's interface for the solve path only):

* :mod:`.configs`            -- ``MpcConfiguration`` / ``CircularRobotSpecification`` from the unchanged yaml keys
                                (reference ``src/configs.py:86-103,140-176``)
* :mod:`.solver`             -- ``solver().run(p, ...)`` object of the generated module (``trajectory_tracker.py:13-15,
                                54-66, 362``) and the batched front end ``BatchSolver``
* :mod:`.trajectory_tracker` -- ``TrajectoryTracker.run_step`` (``trajectory_tracker.py:18-416``)
* :mod:`.mpc_interface`      -- ``MpcInterface.run_step`` (``interfaces/mpc_interface.py:20-102``), static-obstacle
                                marshalling on the device
* :mod:`.solver_build`       -- analogue of ``src/solver_build.py``: compiles the HIP library and writes the
                                ``mpc_solver/<optimizer_name>/`` module the reference imports
* :mod:`.evaluate`           -- ``BatchEvaluator``: the closed-loop evaluation of ``main_base.py:267-346, 448-464`` for B
                                scenarios in lock-step on the device (row f3)
* :mod:`.tcp`                -- OpEn's TCP/JSON wire format in front of the solver + the ``OptimizerTcpManager`` surface
                                used by ``TrajectoryTracker(use_tcp=True)`` (row f4)
* :mod:`.scenarios`          -- synthetic parameter batches of BASELINE.json's configurations
* :mod:`.sharding`           -- one process per GPU, contiguous batch shards, RCCL gather of the results

Device side: ``csrc/`` (hand-written HIP for gfx950) behind the C ABI of ``include/nmpc_hip.h``.
Import of this package never touches the GPU; the library is loaded on first use.
"""
from . import scenarios  # noqa: F401
from ._capi import (EXIT_STATUS_NAMES, EXPORTED_SYMBOLS, Handle, NmpcConfigStruct, NmpcError,  # noqa: F401
                    default_config_struct, layout_info, library_path, load_library)
from .build import build as build_library  # noqa: F401
from .solver import BatchSolver, OptimizerSolution, Solver, make_config, solver  # noqa: F401

__all__ = ["BatchSolver", "OptimizerSolution", "Solver", "make_config", "solver", "scenarios", "Handle", "NmpcConfigStruct", "NmpcError", "default_config_struct", "layout_info", "load_library",
           "library_path", "build_library", "EXIT_STATUS_NAMES", "EXPORTED_SYMBOLS"]
