#!/usr/bin/env python3
"""Diagnostic (needs build/libnmpc_prof.so built with -DNMPC_PROFILE): share of wave cycles per section."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NMPC_HIP_LIBRARY"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "libnmpc_prof.so")
import numpy as np
import dyobav_mpcnwta_warehouse_amd as nm
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = nm.default_config_struct(); cfg.lbfgs_memory = int(os.environ.get("LBFGS_MEM", "10")); cfg.latency_waves = 1; h = nm.Handle(cfg)
L = nm.scenarios.ParamLayout()
P = nm.scenarios.make_batch(B, L, seed=0).astype(np.float32)
U = np.empty((B, 40), np.float32); info = np.empty((B, 24), np.float32)
h.solve_raw(np.float32, P, B, U, info=info)
h.solve_raw(np.float32, P, B, U, info=info)
print("kernel ms", h.last_kernel_ms())
prof = info[:, 8:].astype(np.float64)
names = ["solver (rest: request -> eval entry)", "rollout scans+sincos", "polygons+fleet", "segments+groupmin", "ellipse slots", "pad+control+cost-sum", "adjoint",
         "solver: eval exit -> phase code", "solver: Lipschitz test + L-BFGS update", "solver: two-loop recursion", "solver: line-search test", "solver: step head"]
tot = prof[:, :12].sum()
ne = info[:, 4].astype(np.float64).sum(); ng = info[:, 5].astype(np.float64).sum()
print(f"evals {ne:.3e} (grad {ng:.3e}); cycles/eval total {tot/ne:.0f}")
for i, n in enumerate(names):
    print(f"  {n:28s} {prof[:, i].sum()/tot*100:5.1f}%   {prof[:, i].sum()/ne:8.0f} cycles/eval")
