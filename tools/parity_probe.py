#!/usr/bin/env python3
"""Iterate-path parity probe (GPU fp64 vs CPU oracle) for two scenario families, two Lipschitz-estimator steps and
growing iteration caps: the measurement behind DESIGN.md's "parity protocol" (with OpEn's step 1e-12 the first step
length already differs by ~1e-3 between evaluation orders; with 1e-6 / 1e-4 the paths agree to 1e-9 / 1e-12).
Writes gpurun_out/path_probe2.txt."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dyobav_mpcnwta_warehouse_amd as nm
import oracle
out=[]
def P(*a):
    s = " ".join(str(x) for x in a); print(s, flush=True); out.append(s)
L = nm.scenarios.ParamLayout(); pr = oracle.Problem()
for fam, kw in (("free", dict(n_ped=0, n_boxes=0)), ("oncoming", dict(ped_mode="oncoming"))):
  Pb = nm.scenarios.make_batch(64, L, seed=3, **kw)
  for ld in (1e-6, 1e-4):
    for mo, mi in ((1,1),(1,2),(1,3),(1,5),(1,10),(1,20),(1,50),(2,50),(3,100),(10,500)):
        cfg = nm.default_config_struct(); cfg.max_outer_iterations = mo; cfg.max_inner_iterations = mi
        cfg.lip_delta_f64 = ld; cfg.lip_eps_f64 = ld
        h = nm.Handle(cfg)
        op = oracle.Options(max_outer=mo, max_inner=mi, lip_delta=ld, lip_eps=ld)
        Uo, ro = oracle.solve_batch(pr, op, Pb, nthreads=64)
        r = h.solve(Pb)
        du = np.abs(r["U"] - Uo).max(axis=1)
        P(f"[{fam}] lip {ld} outer {mo} inner {mi}: max|du| {du.max():.3e} med {np.median(du):.3e} n>1e-9 {(du>1e-9).sum()}; inner equal {np.mean(r['iters'][:,1]==ro['inner_iters']):.3f} status equal {np.mean(r['status']==ro['status']):.3f} evals gpu {r['info'][:,4].mean():.1f}/{r['info'][:,5].mean():.1f} orc {ro['n_cost_evals'].mean():.1f}/{ro['n_grad_evals'].mean():.1f}")
        h.close()
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/path_probe2.txt","w").write("\n".join(out)+"\n")
