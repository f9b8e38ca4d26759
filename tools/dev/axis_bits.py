#!/usr/bin/env python3
"""Do the axis-aligned and the general member of a register-table kernel pair give the same bits on axis-aligned ellipses?
(They are chosen per BATCH under axis_aligned = 0: one rotated ellipse anywhere sends every instance to the general member.)
usage: axis_bits.py   -- configs[1] / configs[2] dimensions, two families, fp32 and fp64, throughput and latency kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import dyobav_mpcnwta_warehouse_amd as nm
for wl, key, B in (("cfg1", "cfg1_b1024_n20_2x5", 512), ("cfg2", "cfg2_b65536_n20_4x10", 512)):
    for fam in ("toward_robot", "passing"):
        spec = dict(nm.scenarios.BENCH_CONFIGS[key]); lay = spec.pop("layout"); spec.pop("B")
        P = nm.scenarios.make_batch_chunked(B, lay, ped_mode=fam, dtype=np.float64, **spec)
        for dt in (np.float32, np.float64):
            for lw in (1, 4):
                res = {}
                for ax in (1, -1):
                    cfg = nm.default_config_struct()
                    cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs = lay.N, lay.Nother, lay.Nstc, lay.Ndyn
                    cfg.max_active_dynobs = spec["n_ped"] * spec["n_hyp"]
                    cfg.latency_waves, cfg.staged, cfg.axis_aligned, cfg.tail_latency = lw, -1, ax, -1
                    if dt == np.float64: cfg.reg_table = 1
                    with nm.Handle(cfg) as h:
                        res[ax] = h.solve(P.astype(dt))
                same_u = np.all(res[1]["U"] == res[-1]["U"], axis=1)
                same_it = np.all(res[1]["iters"] == res[-1]["iters"], axis=1)
                du = np.abs(res[1]["U"] - res[-1]["U"]).max(axis=1)
                print(f"{wl} {fam:12s} {np.dtype(dt).name} lw {lw}: axis member vs general member: identical controls {same_u.mean():6.1%}  "
                      f"identical iteration counts {same_it.mean():6.1%}  median |du| {np.median(du):.1e}")
