#!/usr/bin/env python3
"""Diagnostic: solves/s of one BASELINE workload from the kernel time alone (no CPU baseline, no accuracy section).
   usage: quick_rate.py [cfg1|cfg2|cfg4] [B] [launches]   env: LW (latency_waves), COOP (coop_waves), RT (reg_table), DT=f64,
   AX (axis_aligned), STAGED (staged), STAGED2 (staged_evals), POLISH=1, FAMILY, ORDER=lpt, TAIL (tail_latency), BUDGET (max_evaluations)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dyobav_mpcnwta_warehouse_amd as nm
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
key = {"cfg1": "cfg1_b1024_n20_2x5", "cfg2": "cfg2_b65536_n20_4x10", "cfg4": "cfg4_b8192_n40_8x20"}[wl]
spec = nm.scenarios.BENCH_CONFIGS[key]
B = int(sys.argv[2]) if len(sys.argv) > 2 else spec["B"]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dt = np.float64 if os.environ.get("DT") == "f64" else np.float32
spec2 = dict(spec); lay = spec2.pop("layout"); spec2.pop("B")
L, P = lay, nm.scenarios.make_batch_chunked(B, lay, ped_mode=os.environ.get("FAMILY", "toward_robot"), dtype=dt, **spec2)
cfg = nm.default_config_struct()
cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs = L.N, L.Nother, L.Nstc, L.Ndyn
cfg.max_active_dynobs = spec["n_ped"] * spec["n_hyp"]
cfg.latency_waves = int(os.environ.get("LW", "0")); cfg.coop_waves = int(os.environ.get("COOP", "0")); cfg.reg_table = int(os.environ.get("RT", "0"))
cfg.axis_aligned = int(os.environ.get("AX", "0")); cfg.staged = int(os.environ.get("STAGED", "0")); cfg.staged_evals = int(os.environ.get("STAGED2", "0")); cfg.polish = int(os.environ.get("POLISH", "0")); cfg.tail_latency = int(os.environ.get("TAIL", "0")); cfg.max_evaluations = int(os.environ.get("BUDGET", "0"))
h = nm.Handle(cfg)
P = np.ascontiguousarray(P, dtype=dt)
U = np.empty((B, 2 * L.N), dt); it = np.empty((B, 2), np.int32); st = np.empty(B, np.int32)
ms = []
for _ in range(reps + 1):
    h.solve_raw(dt, P, B, U, status=st, iters=it)
    ms.append(h.last_kernel_ms())
if os.environ.get("ORDER") == "lpt":   # longest first, by the evaluation counts of the pass above
    info = np.empty((B, 8), dt); h.solve_raw(dt, P, B, U, status=st, iters=it, info=info)
    h.set_dispatch_order(np.argsort(-info[:, 4], kind="stable").astype(np.int32))
    ms = []
    for _ in range(reps + 1):
        h.solve_raw(dt, P, B, U, status=st, iters=it)
        ms.append(h.last_kernel_ms())
k = float(np.mean(ms[1:]))
print(f"{h.last_launch_info()} ", end="")
print(f"{wl} B={B} {dt.__name__}: kernel {k:.1f} ms -> {B / k * 1e3:.0f} solves/s; inner iters mean {it[:, 1].mean():.0f}; converged {np.mean(st == 0):.4f}; U checksum {float(np.abs(U).sum()):.6f}")
