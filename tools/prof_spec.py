#!/usr/bin/env python3
"""Diagnostic (needs build/libnmpc_prof.so, -DNMPC_PROFILE): cycle shares of the latency kernel's sections for the slowest
instance of the cfg1 batch alone on the chip.  usage: prof_spec.py [W]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NMPC_HIP_LIBRARY"] = os.environ.get("PROF_LIB") or os.path.join(ROOT, "build", "libnmpc_prof.so")
import numpy as np
import dyobav_mpcnwta_warehouse_amd as nm
W = int(sys.argv[1]) if len(sys.argv) > 1 else 3
P = nm.scenarios.make_batch(1024, seed=0).astype(np.float32)[709:710]
cfg = nm.default_config_struct(); cfg.max_active_dynobs = 10; cfg.latency_waves = W
h = nm.Handle(cfg)
U = np.empty((1, 40), np.float32); info = np.empty((1, 24), np.float32); it = np.empty((1, 2), np.int32)
for _ in range(2):
    h.solve_raw(np.float32, P, 1, U, iters=it, info=info)
ms = h.last_kernel_ms()
prof = info[0, 8:].astype(np.float64); n_it = int(it[0, 1])
names = {0: "solver -> eval entry", 1: "rollout", 2: "polygons+fleet", 3: "segments", 4: "ellipses", 5: "pad+control+cost", 6: "adjoint",
         7: "eval epilogue", 8: "exchange (LDS + barrier)", 9: "Lipschitz test", 10: "phase code (candidate replay)", 11: "step head",
         12: "L-BFGS update", 13: "L-BFGS direction (two-loop recursion / compact form)"}
print(f"W={W}: {ms:.2f} ms, {n_it} inner iterations, {ms*1e-3*2.4e9/n_it:.0f} cycles/iter (wavefront 0's stamps below, ticks per iteration)")
names[14] = "requests formed + barrier (A) [then: loop head + evaluation prologue = slot 0]"
names[15] = "round without an acceptable candidate: replay + next requests + barrier (A)"
tot = prof[:16].sum()
for i in range(16):
    print(f"  {names[i]:34s} {prof[i]/n_it:8.0f}  {prof[i]/tot*100:5.1f}%")
