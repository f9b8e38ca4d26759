#!/usr/bin/env python3
"""Diagnostic: per-instance psi-evaluation counts at every outer-iteration boundary (solves are deterministic, so a run
capped at k outer iterations is the first k outer iterations of the full solve). Feeds tools/sim_stages.py, the
list-scheduler replay used to choose the staging policy of the resumable solve.
   usage: dump_outer_profile.py <cfg1|cfg2|cfg4> <family> <B> <out.npz>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import dyobav_mpcnwta_warehouse_amd as nm

wl, family, B, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
key = {"cfg1": "cfg1_b1024_n20_2x5", "cfg2": "cfg2_b65536_n20_4x10", "cfg4": "cfg4_b8192_n40_8x20"}[wl]
spec = dict(nm.scenarios.BENCH_CONFIGS[key])
lay = spec.pop("layout")
spec.pop("B")
P = nm.scenarios.make_batch_chunked(B, lay, ped_mode=family, dtype=np.float32, **spec)
evals = np.zeros((B, 10), np.int32)
f2 = np.zeros((B, 10), np.float32)
status = np.zeros((B, 10), np.int8)
outer = np.zeros((B, 10), np.int8)
ms = []
for k in range(1, 11):
    cfg = nm.default_config_struct()
    cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs = lay.N, lay.Nother, lay.Nstc, lay.Ndyn
    cfg.max_active_dynobs = spec["n_ped"] * spec["n_hyp"]
    cfg.max_outer_iterations = k
    cfg.latency_waves = 1
    with nm.Handle(cfg) as h:
        r = h.solve(P)
        ms.append(h.last_kernel_ms())
    evals[:, k - 1] = r["info"][:, 4].astype(np.int32)
    f2[:, k - 1] = r["info"][:, 1]
    status[:, k - 1] = r["status"]
    outer[:, k - 1] = r["iters"][:, 0]
    print(k, ms[-1], float((r["status"] == 0).mean()), flush=True)
np.savez_compressed(out, evals=evals, f2=f2, status=status, outer=outer, ms=np.array(ms))
