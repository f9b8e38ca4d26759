import json, os, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dyobav_mpcnwta_warehouse_amd as nm
key = "cfg2_b65536_n20_4x10"
spec = dict(nm.scenarios.BENCH_CONFIGS[key]); lay = spec.pop("layout"); spec.pop("B"); spec.pop("seed")
n = 2048
P = nm.scenarios.make_batch(n, lay, seed=1234, ped_mode="passing", **spec)
def cfg_for(**ov):
    cfg = nm.default_config_struct()
    cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs = lay.N, lay.Nother, lay.Nstc, lay.Ndyn
    cfg.max_active_dynobs = spec["n_ped"] * spec["n_hyp"]
    for k, v in ov.items():
        assert hasattr(cfg, k), k
        setattr(cfg, k, v)
    return cfg
def solve(dtype, **ov):
    with nm.Handle(cfg_for(**ov)) as h:
        r = h.solve(P.astype(dtype), dtype=dtype); r["ms"] = h.last_kernel_ms()
    return r
tight = solve(np.float64, tolerance=1e-8, initial_tolerance=1e-8, delta_tolerance=1e-8, max_inner_iterations=2000, max_outer_iterations=15)
plain = solve(np.float32)
t_ok = tight["status"] == 0
du = lambda a, b: np.abs(a.astype(np.float64) - b.astype(np.float64)).max(axis=1)
def q(x): return {"n": int(len(x)), "median": float(np.median(x)), "lt1e-4": float(np.mean(x < 1e-4))}
print(json.dumps({"plain_ms": plain["ms"], "conv": float((plain["status"] == 0).mean())}), flush=True)
for cap in (300, 200, 150, 100, 80, 60, 40):
    r = solve(np.float32, polish=1, polish_max_inner_iterations=cap)
    f = r["info"][:, 6]
    print(json.dumps({"cap": cap, "ms": round(r["ms"], 2), "replaced": int((f == 1).sum()), "selected": int((f >= 1).sum()),
                      "extra_evals_per_selected": round(float((r["info"][:, 4] - plain["info"][:, 4])[f >= 1].mean()), 1),
                      "polished_vs_tight": q(du(r["U"], tight["U"])[(f == 1) & t_ok]), "all_selected_vs_tight": q(du(r["U"], tight["U"])[(f >= 1) & t_ok])}), flush=True)
