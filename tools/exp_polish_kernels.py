#!/usr/bin/env python3
"""Round 6, VERDICT r5 item 5: which fp64 kernel should run the polish's compact batch (and fp64 batches in general) at the
shipped yaml's dimensions -- 13..18 provisioned obstacle rows -- and what the polish costs there.

  a) fp64 solves of configs[1]'s dimensions (Ndynobs = 15) WITHOUT the capacity hint (15 rows provisioned), family `passing`,
     B = 4096 / 32768: the 14-slot fp64 register-table kernel (one wavefront per SIMD: 488-512 VGPRs) against the LDS-table
     kernel (256 VGPRs: two per SIMD) against the automatic choice;
  b) fp32 + polish at the same dimensions, B = 65536, with and without the hint: cost of the polish in per cent.

One JSON line per measurement. usage: exp_polish_kernels.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dyobav_mpcnwta_warehouse_amd as nm

key = "cfg1_b1024_n20_2x5"
spec = dict(nm.scenarios.BENCH_CONFIGS[key]); lay = spec.pop("layout"); spec.pop("B")


def run(P, dt, reps=2, **ov):
    cfg = nm.default_config_struct()
    cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs = lay.N, lay.Nother, lay.Nstc, lay.Ndyn
    for k, v in ov.items():
        setattr(cfg, k, v)
    B = P.shape[0]
    P = np.ascontiguousarray(P, dtype=dt)
    U = np.empty((B, 2 * lay.N), dt); st = np.empty(B, np.int32); info = np.empty((B, 8), dt)
    with nm.Handle(cfg) as h:
        ms = []
        for _ in range(reps + 1):
            h.solve_raw(dt, P, B, U, status=st, info=info)
            ms.append(h.last_kernel_ms())
        li = h.last_launch_info()
    return float(np.mean(ms[1:])), li, float(np.mean(st == 0)), float(info[:, 4].mean())


for B in (4096, 32768):
    P = nm.scenarios.make_batch_chunked(B, lay, ped_mode="passing", dtype=np.float64, **spec)
    for name, ov in (("automatic", {}), ("register table (14 slots, 1 wavefront / SIMD)", dict(reg_table=1, latency_waves=1)),
                     ("LDS table (2 wavefronts / SIMD)", dict(reg_table=-1, latency_waves=1)),
                     ("LDS table, hint 10 rows", dict(reg_table=-1, latency_waves=1, max_active_dynobs=10))):
        ms, li, conv, ev = run(P, np.float64, **ov)
        print(json.dumps({"exp": "a", "B": B, "dtype": "f64", "kernel": name, "kernel_ms": round(ms, 2), "solves_per_s": round(B / ms * 1e3),
                          "family": li["family"], "converged": round(conv, 3), "psi_evals_per_solve": round(ev, 1)}), flush=True)
B = 65536
P = nm.scenarios.make_batch_chunked(B, lay, ped_mode="passing", dtype=np.float32, **spec)
for hint in (10, 0):
    base = run(P, np.float32, max_active_dynobs=hint)
    pol = run(P, np.float32, max_active_dynobs=hint, polish=1)
    print(json.dumps({"exp": "b", "B": B, "max_active_dynobs": hint, "kernel_ms": round(base[0], 2), "kernel_ms_with_polish": round(pol[0], 2),
                      "polish_cost_percent": round(100 * (pol[0] / base[0] - 1), 1), "polish_selected": pol[1]["polish_selected"],
                      "solves_per_s": round(B / base[0] * 1e3), "solves_per_s_with_polish": round(B / pol[0] * 1e3)}), flush=True)
