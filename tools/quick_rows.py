#!/usr/bin/env python3
"""A few bench rows in one process (headline, `passing`, polish, configs[1]) -- for A/B runs during kernel work.
usage: quick_rows.py [steps]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

class A: gpus = 1
env = bench.Env(A())
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rows = [("cfg2", "toward_robot", "f32", steps, 1, {}), ("cfg2", "passing", "f32", steps, 1, {}),
        ("cfg2", "passing", "f32", steps, 1, {"polish": True}), ("cfg1", "toward_robot", "f32", 5, 1, {}),
        ("cfg1", "passing", "f32", 5, 1, {}), ("cfg1", "toward_robot", "f32", 2, 1, {"batch": 65536}),
        ("cfg4", "toward_robot", "f32", 1, 1, {}), ("cfg2", "toward_robot", "f64", 1, 0, {"batch": 16384})]
for wl, fam, dt, st, wu, kw in rows:
    r = bench.run_workload(env, wl, fam, dt, st, wu, **kw)
    print(json.dumps({"row": f"{wl} {fam} {dt} {kw}", "solves_per_s": round(r["value"]), "kernel_ms": round(r["roofline"]["kernel_ms"], 2),
                      "evals": round(r["roofline"]["psi_evals_per_solve"], 1), "valu_frac": round(r["roofline"]["valu_frac"], 4),
                      "converged": round(r["solver"]["converged_frac"], 4), "polish": r["polish"]}), flush=True)
