#!/usr/bin/env python3
"""Experiment (round 3): does an fp64 continuation ("polish") of a converged fp32 solve land on the fixed point?

Prototype over the EXISTING C ABI: fp32 solve -> instances flagged Converged -> fp64 solve warm-started from
(u, y, c) of the fp32 result with tightened tolerances. Compared with fp64 solves from scratch at the default and at
tight tolerance. Prints one JSON record per (workload, polish tolerance).
   usage: exp_polish.py [cfg1|cfg2] [n] [family]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import dyobav_mpcnwta_warehouse_amd as nm

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
family = sys.argv[3] if len(sys.argv) > 3 else "passing"
key = {"cfg1": "cfg1_b1024_n20_2x5", "cfg2": "cfg2_b65536_n20_4x10", "cfg4": "cfg4_b8192_n40_8x20"}[wl]
spec = dict(nm.scenarios.BENCH_CONFIGS[key])
lay = spec.pop("layout")
spec.pop("B")
spec.pop("seed")
P = nm.scenarios.make_batch(n, lay, seed=1234, ped_mode=family, **spec)
LIP = 1e-4


def cfg_for(**ov):
    cfg = nm.default_config_struct()
    cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs = lay.N, lay.Nother, lay.Nstc, lay.Ndyn
    cfg.max_active_dynobs = spec["n_ped"] * spec["n_hyp"]
    cfg.lip_eps_f64 = cfg.lip_delta_f64 = cfg.lip_eps_f32 = cfg.lip_delta_f32 = LIP
    for k, v in ov.items():
        assert hasattr(cfg, k), k
        setattr(cfg, k, v)
    return cfg


def solve(dtype, Pb, **kw):
    ov = {k: kw.pop(k) for k in list(kw) if k not in ("u0", "y0", "c0")}
    with nm.Handle(cfg_for(**ov)) as h:
        r = h.solve(Pb.astype(dtype), dtype=dtype, **kw)
        r["ms"] = h.last_kernel_ms()
    return r


def du(a, b):
    return np.abs(a.astype(np.float64) - b.astype(np.float64)).max(axis=1)


def q(x):
    if len(x) == 0:
        return None
    return {"n": int(len(x)), "median": float(np.median(x)), "q90": float(np.quantile(x, 0.9)), "max": float(x.max()),
            "frac_lt_1e-4": float(np.mean(x < 1e-4))}


r32 = solve(np.float32, P)
r64 = solve(np.float64, P)
tight = dict(tolerance=1e-8, initial_tolerance=1e-8, delta_tolerance=1e-8, max_inner_iterations=2000, max_outer_iterations=15)
r64t = solve(np.float64, P, **tight)
c32, c64 = r32["status"] == 0, r64["status"] == 0
print(json.dumps({"workload": wl, "family": family, "n": n, "conv32": float(c32.mean()), "conv64": float(c64.mean()),
                  "conv64_tight": float((r64t["status"] == 0).mean()),
                  "ms32": r32["ms"], "ms64": r64["ms"], "ms64_tight": r64t["ms"],
                  "evals32": float(r32["info"][:, 4].mean()), "evals64": float(r64["info"][:, 4].mean()),
                  "hip32_vs_hip64(both conv)": q(du(r32["U"], r64["U"])[c32 & c64]),
                  "hip64_vs_hip64tight(both conv)": q(du(r64["U"], r64t["U"])[c64 & (r64t["status"] == 0)]),
                  "hip32_vs_hip64tight(both conv)": q(du(r32["U"], r64t["U"])[c32 & (r64t["status"] == 0)])}), flush=True)


def polish(r, sel, tol, delta, max_outer, max_inner, keep_c=True):
    idx = np.flatnonzero(sel)
    kw = dict(tolerance=tol, initial_tolerance=tol, delta_tolerance=delta, max_inner_iterations=max_inner,
              max_outer_iterations=max_outer)
    rp = solve(np.float64, P[idx], u0=r["U"][idx].astype(np.float64), y0=r["y"][idx].astype(np.float64),
               c0=(r["info"][idx, 3].astype(np.float64) if keep_c else None), **kw)
    return idx, rp


for tol, delta, mo, mi in ((1e-6, 1e-6, 4, 300), (1e-7, 1e-7, 6, 500), (1e-8, 1e-8, 8, 1000), (1e-6, 1e-4, 4, 300),
                           (1e-7, 1e-4, 4, 500)):
    i32, p32 = polish(r32, c32, tol, delta, mo, mi)
    i64, p64 = polish(r64, c64, tol, delta, mo, mi)
    U32p, U64p = r32["U"].astype(np.float64).copy(), r64["U"].copy()
    ok32 = np.zeros(n, bool)
    ok64 = np.zeros(n, bool)
    ok32[i32] = p32["status"] == 0
    ok64[i64] = p64["status"] == 0
    U32p[i32] = p32["U"]
    U64p[i64] = p64["U"]
    t_ok = r64t["status"] == 0
    rec = {"polish": {"tol": tol, "delta": delta, "max_outer": mo, "max_inner": mi},
           "n_polished32": int(len(i32)), "polish_conv32": float(np.mean(p32["status"] == 0)) if len(i32) else None,
           "polish_conv64": float(np.mean(p64["status"] == 0)) if len(i64) else None,
           "polish32_ms": p32["ms"], "polish32_evals_mean": float(p32["info"][:, 4].mean()),
           "polish32_evals_max": float(p32["info"][:, 4].max()),
           "polish32_outer_mean": float(p32["iters"][:, 0].mean()), "polish32_inner_mean": float(p32["iters"][:, 1].mean()),
           "hip32p_vs_hip64p(both polished ok)": q(du(U32p, U64p)[ok32 & ok64]),
           "hip32p_vs_hip64p(both polished, any status)": q(du(U32p, U64p)[c32 & c64]),
           "hip32p_vs_hip64tight(ok & tight conv)": q(du(U32p, r64t["U"])[ok32 & t_ok]),
           "hip64p_vs_hip64tight(ok & tight conv)": q(du(U64p, r64t["U"])[ok64 & t_ok]),
           "hip32p_moved": q(du(U32p, r32["U"])[ok32])}
    print(json.dumps(rec), flush=True)
