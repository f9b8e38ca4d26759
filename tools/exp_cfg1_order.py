#!/usr/bin/env python3
"""Experiment: does a dispatch order ranked by what a one-outer-iteration pilot finds shorten the configs[1] launch
(all 1024 workgroups resident at once: the order decides which instances share a SIMD)?"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dyobav_mpcnwta_warehouse_amd as nm
spec = dict(nm.scenarios.BENCH_CONFIGS["cfg1_b1024_n20_2x5"]); lay = spec.pop("layout"); B = spec.pop("B")
fam = sys.argv[1] if len(sys.argv) > 1 else "toward_robot"
P = nm.scenarios.make_batch_chunked(B, lay, ped_mode=fam, dtype=np.float32, **spec)
def mk(**ov):
    cfg = nm.default_config_struct(); cfg.max_active_dynobs = 10
    for k, v in ov.items(): setattr(cfg, k, v)
    return nm.Handle(cfg)
def timed(h, n=5):
    ms = []
    for _ in range(n):
        r = h.solve(P); ms.append(h.last_kernel_ms())
    return r, float(np.median(ms[1:]))
for k in (1, 2):
    pil, t_p = timed(mk(max_outer_iterations=k))
    h = mk()
    full, t_idx = timed(h)
    out = {"family": fam, "pilot_outer": k, "pilot_ms": t_p, "index_ms": t_idx}
    for name, key in (("f2", -pil["info"][:, 1]), ("pilot evals", -pil["info"][:, 4]), ("fpr", -pil["info"][:, 0]), ("perfect", -full["info"][:, 4]),
                      ("f2 then evals", -1e6 * (pil["info"][:, 1] > 1e-4) - pil["info"][:, 4])):
        h.set_dispatch_order(np.argsort(key, kind="stable").astype(np.int32))
        _, t = timed(h)
        out[name + "_ms"] = t
    for w in (4,):
        h4 = mk(latency_waves=w)
        _, t4 = timed(h4)
        out[f"W{w}_index_ms"] = t4
    print(out, flush=True)
