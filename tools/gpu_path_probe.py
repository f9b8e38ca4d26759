#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dyobav_mpcnwta_warehouse_amd as nm
import oracle
out=[]
def P(*a):
    s = " ".join(str(x) for x in a); print(s, flush=True); out.append(s)
L = nm.scenarios.ParamLayout(); pr = oracle.Problem()
Pb = nm.scenarios.make_batch(64, L, seed=3, ped_mode="oncoming")
for mo, mi in ((1,1),(1,2),(1,3),(1,5),(1,10),(1,20),(1,50),(1,200),(2,50),(3,50),(5,100)):
    cfg = nm.default_config_struct(); cfg.max_outer_iterations = mo; cfg.max_inner_iterations = mi
    h = nm.Handle(cfg)
    op = oracle.Options(max_outer=mo, max_inner=mi)
    Uo, ro = oracle.solve_batch(pr, op, Pb, nthreads=64)
    r = h.solve(Pb)
    du = np.abs(r["U"] - Uo).max(axis=1)
    P(f"outer {mo} inner {mi}: max|du| {du.max():.3e} med {np.median(du):.3e} n>1e-9 {(du>1e-9).sum()}; inner iters equal {np.mean(r['iters'][:,1]==ro['inner_iters']):.3f} status equal {np.mean(r['status']==ro['status']):.3f} cost rel {np.max(np.abs(r['cost']-ro['cost'])/np.abs(ro['cost'])):.2e} dy {np.max(np.abs(r['info'][:,2]-ro['delta_y_norm'])):.2e} c eq {np.mean(r['info'][:,3]==ro['penalty']):.2f}")
    h.close()
# tight tolerance
for tol in (1e-6, 1e-8):
    cfg = nm.default_config_struct(); cfg.tolerance = tol; cfg.initial_tolerance = tol; cfg.delta_tolerance = tol; cfg.max_inner_iterations = 5000; cfg.max_outer_iterations = 30
    h = nm.Handle(cfg)
    op = oracle.Options(tolerance=tol, initial_tolerance=tol, delta_tolerance=tol, max_inner=5000, max_outer=30)
    for fam, kw in (("free", dict(n_ped=0, n_boxes=0)), ("boxes", dict(n_ped=0)), ("oncoming", dict(ped_mode="oncoming"))):
        Pb2 = nm.scenarios.make_batch(64, L, seed=5, **kw)
        Uo, ro = oracle.solve_batch(pr, op, Pb2, nthreads=64)
        r = h.solve(Pb2)
        both = (r["status"]==0)&(ro["status"]==0)
        du = np.abs(r["U"] - Uo).max(axis=1)
        P(f"tol {tol} [{fam}]: oracle conv {np.mean(ro['status']==0):.2f} gpu conv {np.mean(r['status']==0):.2f}; both n={both.sum()} du max {du[both].max() if both.any() else -1:.2e} med {np.median(du[both]) if both.any() else -1:.2e}; inner mean {r['iters'][:,1].mean():.0f}/{ro['inner_iters'].mean():.0f} outer mean {r['iters'][:,0].mean():.1f}")
        r32 = h.solve(Pb2.astype(np.float32))
        b32 = (r32["status"]==0)&(ro["status"]==0)
        du32 = np.abs(r32["U"].astype(np.float64) - Uo).max(axis=1)
        P(f"      f32: conv {np.mean(r32['status']==0):.2f} both n={b32.sum()} du max {du32[b32].max() if b32.any() else -1:.2e} med {np.median(du32[b32]) if b32.any() else -1:.2e}; all: med {np.median(du32):.2e}")
    h.close()
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/path_probe.txt","w").write("\n".join(out)+"\n")
