#!/usr/bin/env python3
"""Host-side cost of the B = 1 drop-in path: wall time of `solver().run(p)` (a Python list in, OptimizerSolution out)
against the HIP-event time of the kernel it launches -- the path main.py uses once per control step."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dyobav_mpcnwta_warehouse_amd as nm
from dyobav_mpcnwta_warehouse_amd.solver import Solver

P = nm.scenarios.make_batch(64, seed=5, ped_mode="passing")
s = Solver(nm.default_config_struct(), dtype=np.float64, keep_multipliers=False)
rows = [[float(v) for v in p] for p in P]
for r in rows[:4]:
    s.run(r)
wall, kern = [], []
for r in rows:
    t0 = time.perf_counter()
    sol = s.run(r)
    wall.append((time.perf_counter() - t0) * 1e3)
    kern.append(s._handle.last_kernel_ms())
wall, kern = np.array(wall), np.array(kern)
print(json.dumps({"metric": "B=1 solver().run(p) latency", "n": len(rows), "wall_ms_median": float(np.median(wall)),
                  "kernel_ms_median": float(np.median(kern)), "host_ms_median": float(np.median(wall - kern)),
                  "host_ms_p90": float(np.quantile(wall - kern, 0.9)), "dtype": "f64"}))
