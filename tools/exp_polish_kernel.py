#!/usr/bin/env python3
"""Experiment: which fp64 kernel should run the polish batch (converged instances of configs[2] `passing`, warm-started,
tolerance 1e-6, 4 x 300 iterations)? Times the fp64 continuation alone under different kernel choices."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dyobav_mpcnwta_warehouse_amd as nm
spec = dict(nm.scenarios.BENCH_CONFIGS["cfg2_b65536_n20_4x10"]); lay = spec.pop("layout"); spec.pop("B")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
P = nm.scenarios.make_batch_chunked(B, lay, ped_mode="passing", dtype=np.float32, **spec)
def cfg_for(**ov):
    cfg = nm.default_config_struct()
    cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs = lay.N, lay.Nother, lay.Nstc, lay.Ndyn
    cfg.max_active_dynobs = 40
    for k, v in ov.items(): setattr(cfg, k, v)
    return cfg
with nm.Handle(cfg_for()) as h:
    r = h.solve(P)
sel = np.flatnonzero(r["status"] == 0)
Ps = P[sel].astype(np.float64); u0 = r["U"][sel].astype(np.float64); y0 = r["y"][sel].astype(np.float64); c0 = r["info"][sel, 3].astype(np.float64)
print("selected", len(sel), "of", B)
for name, ov in (("auto", {}), ("throughput (1 wave)", dict(latency_waves=1, coop_waves=1)), ("latency W=2", dict(latency_waves=2)),
                 ("latency W=3", dict(latency_waves=3)), ("latency W=4", dict(latency_waves=4)), ("coop W=4", dict(latency_waves=1, coop_waves=4)),
                 ("coop W=2", dict(latency_waves=1, coop_waves=2)),
                 ("fp64 register table, 1 wave/SIMD", dict(latency_waves=1, coop_waves=1, reg_table=1))):
    with nm.Handle(cfg_for(tolerance=1e-6, initial_tolerance=1e-6, delta_tolerance=1e-5, max_outer_iterations=4, max_inner_iterations=300, staged=-1, **ov)) as h:
        ms = []
        for _ in range(2):
            q = h.solve(Ps, u0=u0, y0=y0, c0=c0, dtype=np.float64); ms.append(h.last_kernel_ms())
        print(f"{name:22s} {ms[-1]:8.1f} ms  family {h.last_launch_info()['family']}  converged {np.mean(q['status'] == 0):.3f}  evals/inst {q['info'][:, 4].mean():.0f}", flush=True)
