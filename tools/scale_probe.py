#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dyobav_mpcnwta_warehouse_amd as nm
Bs = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1024, 2048, 4096, 8192, 16384]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
h = nm.Handle(nm.default_config_struct())
print("kernel_info", h.kernel_info(), flush=True)
L = nm.scenarios.ParamLayout()
for B in Bs:
    P = nm.scenarios.make_batch(B, L, seed=0).astype(np.float32)
    ts = []
    for _ in range(reps):
        r = h.solve(P); ts.append(h.last_kernel_ms())
    ev = r["info"][:, 4].astype(np.float64)
    print(f"B={B}: kernel ms {min(ts):.2f} -> {B/(min(ts)*1e-3):.0f} solves/s; evals/solve mean {ev.mean():.0f} max {ev.max():.0f}; sum evals {ev.sum():.3e} -> {ev.sum()/(min(ts)*1e-3)/1e6:.1f} M evals/s", flush=True)
