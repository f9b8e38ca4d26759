#!/usr/bin/env python3
"""Experiment: configs[1] as a multi-stage solve -- pilot of k outer iterations in index order, the rest re-dispatched in
an order ranked by what the pilot saw. Estimate of the staged time = pilot_k(index) + [full(ranked) - pilot_k(ranked)]."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dyobav_mpcnwta_warehouse_amd as nm
spec = dict(nm.scenarios.BENCH_CONFIGS["cfg1_b1024_n20_2x5"]); lay = spec.pop("layout"); B = spec.pop("B")
P = nm.scenarios.make_batch_chunked(B, lay, ped_mode="toward_robot", dtype=np.float32, **spec)
def mk(**ov):
    cfg = nm.default_config_struct(); cfg.max_active_dynobs = 10
    for k, v in ov.items(): setattr(cfg, k, v)
    return nm.Handle(cfg)
def timed(h, order=None, n=4):
    h.set_dispatch_order(order)
    ms = []
    for _ in range(n):
        r = h.solve(P); ms.append(h.last_kernel_ms())
    return r, float(np.median(ms[1:]))
hf = mk()
full, t_idx = timed(hf)
print("full index", t_idx, "perfect", timed(hf, np.argsort(-full["info"][:, 4], kind="stable").astype(np.int32))[1], flush=True)
prev = None
for k in range(1, 10):
    hp = mk(max_outer_iterations=k)
    pil, t_p = timed(hp)
    ev = pil["info"][:, 4]
    inc = ev if prev is None else ev - prev
    prev = ev
    res = {"k": k, "pilot_index_ms": round(t_p, 2)}
    for name, key in (("cum", -ev), ("inc", -inc), ("f2", -pil["info"][:, 1])):
        o = np.argsort(key, kind="stable").astype(np.int32)
        _, t_full_r = timed(hf, o)
        _, t_pil_r = timed(hp, o)
        res[name] = {"full_ranked": round(t_full_r, 2), "pilot_ranked": round(t_pil_r, 2), "staged_estimate": round(t_p + t_full_r - t_pil_r, 2)}
    print(res, flush=True)
