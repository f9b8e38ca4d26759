#!/usr/bin/env python3
"""Experiment: where the fp64 polish spends its evaluations. The polish leg is emulated through the public API
(run(p, u, y, c) on a tighter fp64 solver) with the outer-iteration cap k = 1..4: evaluations, inner iterations and the
distance to the tolerance-1e-8 fixed point after k outer iterations, for several (tolerance, delta) pairs.
usage: exp_polish_stats.py [cfg1|cfg2] [n]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dyobav_mpcnwta_warehouse_amd as nm
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
key = {"cfg1": "cfg1_b1024_n20_2x5", "cfg2": "cfg2_b65536_n20_4x10"}[wl]
spec = dict(nm.scenarios.BENCH_CONFIGS[key]); lay = spec.pop("layout"); spec.pop("B"); spec.pop("seed")
P = nm.scenarios.make_batch(n, lay, seed=1234, ped_mode="passing", **spec)
def cfg_for(**ov):
    cfg = nm.default_config_struct()
    cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs = lay.N, lay.Nother, lay.Nstc, lay.Ndyn
    cfg.max_active_dynobs = spec["n_ped"] * spec["n_hyp"]
    for k, v in ov.items():
        assert hasattr(cfg, k), k
        setattr(cfg, k, v)
    return cfg
def solve(dtype, Pm=P, **kw):
    ov = {k: v for k, v in kw.items() if k not in ("u0", "y0", "c0")}
    with nm.Handle(cfg_for(**ov)) as h:
        r = h.solve(Pm.astype(dtype), dtype=dtype, u0=kw.get("u0"), y0=kw.get("y0"), c0=kw.get("c0")); r["ms"] = h.last_kernel_ms()
    return r
tight = solve(np.float64, tolerance=1e-8, initial_tolerance=1e-8, delta_tolerance=1e-8, max_inner_iterations=2000, max_outer_iterations=15)
plain = solve(np.float32)
sel = (plain["status"] == 0) & (tight["status"] == 0)
Ps, u0, y0, c0 = P[sel], plain["U"][sel].astype(np.float64), plain["y"][sel].astype(np.float64), plain["info"][sel, 3].astype(np.float64)
ref = tight["U"][sel]
du = lambda a: np.abs(a - ref).max(axis=1)
def q(x):
    return {"median": float(np.median(x)), "q90": float(np.quantile(x, .9)), "lt1e-4": float(np.mean(x < 1e-4))}
print(json.dumps({"workload": wl, "n": n, "selected": int(sel.sum()), "plain_evals": float(plain["info"][sel, 4].mean()), "plain_vs_tight": q(du(u0))}), flush=True)
for tol, delta in ((1e-6, 1e-5), (1e-6, 1e-4), (1e-7, 1e-5), (3e-6, 1e-5), (1e-5, 1e-4)):
    for k in (1, 2, 3, 4):
        r = solve(np.float64, Pm=Ps, u0=u0, y0=y0, c0=c0, tolerance=tol, initial_tolerance=tol, delta_tolerance=delta,
                  max_outer_iterations=k, max_inner_iterations=300, lip_eps_f64=1e-6, lip_delta_f64=1e-12)
        print(json.dumps({"tol": tol, "delta": delta, "max_outer": k, "converged": float((r["status"] == 0).mean()), "evals": float(r["info"][:, 4].mean()),
                          "grad_evals": float(r["info"][:, 5].mean()), "inner": float(r["iters"][:, 1].mean()), "outer": float(r["iters"][:, 0].mean()),
                          "penalty_grew": float((r["info"][:, 3] > c0 * 1.01).mean()), "vs_tight_all": q(du(r["U"])),
                          "vs_tight_converged": q(du(r["U"])[r["status"] == 0]) if (r["status"] == 0).any() else None, "ms": r["ms"]}), flush=True)
