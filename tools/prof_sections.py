#!/usr/bin/env python3
"""Diagnostic (needs build/libnmpc_prof.so built with -DNMPC_PROFILE): share of wave cycles per section.
   usage: prof_sections.py [B] [workload: cfg1|cfg2|cfg4] [family: toward_robot|passing]
   (build: hipcc <the flags of build.py> -DNMPC_PROFILE=1 -o build/libnmpc_prof.so csrc/nmpc_capi.hip; =2 adds the event counters of the passes)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NMPC_HIP_LIBRARY"] = os.path.join(ROOT, "build", "libnmpc_prof.so")
import numpy as np
import dyobav_mpcnwta_warehouse_amd as nm
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
wl = sys.argv[2] if len(sys.argv) > 2 else "cfg1"
key = {"cfg1": "cfg1_b1024_n20_2x5", "cfg2": "cfg2_b65536_n20_4x10", "cfg4": "cfg4_b8192_n40_8x20"}[wl]
fam = sys.argv[3] if len(sys.argv) > 3 else "toward_robot"
spec = dict(nm.scenarios.BENCH_CONFIGS[key])
L = spec.pop("layout"); spec.pop("B")
P = nm.scenarios.make_batch_chunked(B, L, ped_mode=fam, dtype=np.float64, **spec)
cfg = nm.default_config_struct(); cfg.lbfgs_memory = int(os.environ.get("LBFGS_MEM", "10")); cfg.latency_waves = int(os.environ.get("LW", "1"))
cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs = L.N, L.Nother, L.Nstc, L.Ndyn
cfg.max_active_dynobs = spec["n_ped"] * spec["n_hyp"]
cfg.staged = -1                                   # (the parked state does not carry the profile counters)
cfg.axis_aligned = int(os.environ.get("AX", "0"))
h = nm.Handle(cfg)
P = P.astype(np.float32)
U = np.empty((B, 2 * L.N), np.float32); info = np.empty((B, 24), np.float32)
h.solve_raw(np.float32, P, B, U, info=info)
h.solve_raw(np.float32, P, B, U, info=info)
print(wl, fam, "B", B, "kernel ms", h.last_kernel_ms())
prof = info[:, 8:].astype(np.float64)
names = ["solver (rest: request -> eval entry)", "rollout scans+sincos", "polygons+fleet", "segments+groupmin", "ellipse slots", "pad+control+cost-sum", "adjoint",
         "solver: eval exit -> phase code", "solver: Lipschitz test + L-BFGS update", "solver: two-loop recursion", "solver: line-search test", "solver: step head"]
coop = bool((info[:, 7] < 0).any())
tot = prof[:, :14].sum() if coop else prof[:, :12].sum()
ne = info[:, 4].astype(np.float64).sum(); ng = info[:, 5].astype(np.float64).sum()
print(f"evals {ne:.3e} (grad {ng:.3e}) = {ne/B:.0f} per solve; cycles/eval total {tot/ne:.0f}")
for i, n in enumerate(names):
    print(f"  {n:40s} {prof[:, i].sum()/tot*100:5.1f}%   {prof[:, i].sum()/ne:8.0f} ticks/eval")
if coop:
    print(f"  cooperative kernel: register passes {prof[:, 13].sum()/ne:8.0f}, other rows + routing {prof[:, 12].sum()/ne:8.0f}, exchange + barrier {prof[:, 4].sum()/ne:8.0f} ticks/eval")
    sys.exit(0)
if prof[:, 13].sum() > 0:
  print(f"ellipse passes visited {prof[:,13].sum():.3e}; any soft term active {prof[:,14].sum()/prof[:,13].sum()*100:.1f}%; any hard term active {prof[:,15].sum()/prof[:,13].sum()*100:.1f}%; "
      f"per pair of slots")
