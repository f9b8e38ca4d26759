#!/usr/bin/env python3
"""Diagnostic run on the GPU box: wave self-test, psi/grad parity, solve parity and a first timing.
Writes a report to gpurun_out/first_light.txt (scratch, not judged)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dyobav_mpcnwta_warehouse_amd as nm
import oracle

out = []
def P(*a):
    s = " ".join(str(x) for x in a); print(s, flush=True); out.append(s)

cfg = nm.default_config_struct()
h = nm.Handle(cfg)
P("kernel_info", h.kernel_info())
P("selftest failures:", h.selftest())

gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
fx = np.load(os.path.join(gold, "problem_n20.npz"))
pr = oracle.Problem()
rng = np.random.default_rng(0)
K = fx["P"].shape[0]
for dt, tol in ((np.float64, 1e-9), (np.float32, 2e-4)):
    Y = rng.normal(size=(K, 40)); Cp = rng.uniform(1, 100, K)
    Cp[:4] = 0.0
    r = h.eval(fx["P"].astype(dt), fx["U"].astype(dt), Y.astype(dt), Cp.astype(dt), dtype=dt)
    epsi = eg = ef2 = 0
    for i in range(K):
        v, g = oracle.psi(pr, fx["U"][i], Cp[i], Y[i], fx["P"][i])
        epsi = max(epsi, abs(r["psi"][i]-v)/abs(v)); eg = max(eg, np.abs(r["grad"][i]-g).max()/np.abs(g).max())
        ef2 = max(ef2, abs(r["f2sq"][i]-np.sum(fx["F2"][i]**2))/max(1,np.sum(fx["F2"][i]**2)))
    P(f"eval parity {np.dtype(dt).name}: rel psi {epsi:.3e} rel grad {eg:.3e} rel f2sq {ef2:.3e}")

L = nm.scenarios.ParamLayout()
Pb = nm.scenarios.make_batch(64, L, seed=0)
op = oracle.Options()
t = time.time(); Uo, ro = oracle.solve_batch(pr, op, Pb, nthreads=os.cpu_count()); to = time.time()-t
P(f"oracle f64: {to:.2f}s for 64 on {os.cpu_count()} threads; status {np.bincount(ro['status'])} inner mean {ro['inner_iters'].mean():.0f}")
for dt in (np.float64, np.float32):
    t = time.time(); r = h.solve(Pb.astype(dt)); tg = time.time()-t
    du = np.abs(r["U"].astype(np.float64) - Uo).max(axis=1)
    same = (r["status"] == ro["status"])
    P(f"gpu {np.dtype(dt).name}: wall {tg*1e3:.1f} ms kernel {h.last_kernel_ms():.2f} ms; status {np.bincount(r['status'], minlength=4)} same-status {same.mean():.3f}; "
      f"inner mean {r['iters'][:,1].mean():.0f} (oracle {ro['inner_iters'].mean():.0f}); max|du| {du.max():.3e} median {np.median(du):.3e}; "
      f"frac<1e-4 {(du<1e-4).mean():.3f}; cost rel {np.abs(r['cost']-ro['cost']).max()/np.abs(ro['cost']).max():.2e}; evals {r['info'][:,4].mean():.0f}/{r['info'][:,5].mean():.0f}")
    P("   worst instances:", np.argsort(-du)[:5], du[np.argsort(-du)[:5]])

# timing at B=1024 / 8192
for B in (1024, 8192):
    Pb = nm.scenarios.make_batch(B, L, seed=0).astype(np.float32)
    h.solve(Pb)
    ts = []
    for _ in range(3):
        h.solve(Pb); ts.append(h.last_kernel_ms())
    P(f"f32 B={B}: kernel ms {ts} -> {B/(min(ts)*1e-3):.0f} solves/s")
Pb = nm.scenarios.make_batch(1024, L, seed=0)
h.solve(Pb); h.solve(Pb)
P(f"f64 B=1024: kernel ms {h.last_kernel_ms():.2f} -> {1024/(h.last_kernel_ms()*1e-3):.0f} solves/s")
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/first_light.txt", "w").write("\n".join(out) + "\n")
