#!/usr/bin/env python3
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dyobav_mpcnwta_warehouse_amd as nm
import oracle
out=[]
def P(*a):
    s = " ".join(str(x) for x in a); print(s, flush=True); out.append(s)
cfg = nm.default_config_struct()
h = nm.Handle(cfg)
L = nm.scenarios.ParamLayout(); pr = oracle.Problem(); op = oracle.Options()
fams = {"free": dict(n_ped=0, n_boxes=0), "boxes": dict(n_ped=0, n_boxes=4), "oncoming": dict(n_ped=2, n_hyp=5, ped_mode="oncoming"), "toward": dict()}
for name, kw in fams.items():
    Pb = nm.scenarios.make_batch(128, L, seed=3, **kw)
    Uo, ro = oracle.solve_batch(pr, op, Pb, nthreads=64)
    Uo32, ro32 = oracle.solve_batch(pr, oracle.Options(lip_delta=1e-4, lip_eps=1e-4), Pb, nthreads=64, dtype=np.float32)
    P(f"[{name}] oracle f64 status {np.bincount(ro['status'],minlength=2)} outer {np.bincount(ro['outer_iters'])} inner mean {ro['inner_iters'].mean():.0f}; oracle f32 status {np.bincount(ro32['status'],minlength=2)} inner {ro32['inner_iters'].mean():.0f} max|du32-64| conv {np.abs(Uo32-Uo)[(ro['status']==0)&(ro32['status']==0)].max() if ((ro['status']==0)&(ro32['status']==0)).any() else -1:.2e}")
    for dt in (np.float64, np.float32):
        r = h.solve(Pb.astype(dt))
        du = np.abs(r["U"].astype(np.float64) - Uo).max(axis=1)
        both = (r["status"] == 0) & (ro["status"] == 0)
        P(f"   gpu {np.dtype(dt).name}: status {np.bincount(r['status'],minlength=4)} same {np.mean(r['status']==ro['status']):.3f} outer {np.bincount(r['iters'][:,0])} inner {r['iters'][:,1].mean():.0f} same-iters {np.mean((r['iters'][:,1]==ro['inner_iters'])):.3f}; "
          f"du(all) max {du.max():.2e} med {np.median(du):.2e}; du(both conv, n={both.sum()}) max {du[both].max() if both.any() else -1:.2e} med {np.median(du[both]) if both.any() else -1:.2e} frac<1e-4 {(du[both]<1e-4).mean() if both.any() else -1:.3f}")
        if dt == np.float64:
            i = int(np.argmax(np.where(both, du, -1)))
            P(f"      worst conv inst {i}: du {du[i]:.3e} iters gpu {r['iters'][i]} oracle ({ro['outer_iters'][i]},{ro['inner_iters'][i]}) cost gpu {r['cost'][i]:.6f} oracle {ro['cost'][i]:.6f} fpr {r['info'][i,0]:.2e}/{ro['last_fpr'][i]:.2e}")
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/parity_probe.txt","w").write("\n".join(out)+"\n")
