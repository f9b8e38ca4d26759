#!/usr/bin/env python3
"""configs[1], latency kernel: exchange rounds and inner iterations of the slowest instances, for W = 2..4 (results do not
depend on W; the rounds do).   usage: exp_cfg1_rounds.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dyobav_mpcnwta_warehouse_amd as nm
spec = dict(nm.scenarios.BENCH_CONFIGS["cfg1_b1024_n20_2x5"]); lay = spec.pop("layout"); B = spec.pop("B")
P = nm.scenarios.make_batch_chunked(B, lay, ped_mode="toward_robot", dtype=np.float32, **spec)
for W in (2, 3, 4):
    cfg = nm.default_config_struct()
    cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs = lay.N, lay.Nother, lay.Nstc, lay.Ndyn
    cfg.max_active_dynobs = spec["n_ped"] * spec["n_hyp"]; cfg.latency_waves = W; cfg.staged = -1
    with nm.Handle(cfg) as h:
        r = h.solve(P); r = h.solve(P)
        ms = h.last_kernel_ms()
    it = r["iters"][:, 1].astype(np.int64); rounds = r["info"][:, 6].astype(np.int64); ev = r["info"][:, 4].astype(np.int64)
    o = np.argsort(-rounds)[:8]
    print(f"W={W}: kernel {ms:.2f} ms; all instances: iterations {it.sum()}, rounds {rounds.sum()}, sequential evaluations {ev.sum()} -> {rounds.sum()/it.sum():.2f} rounds and {ev.sum()/it.sum():.2f} evaluations per iteration")
    print("   slowest by rounds: " + ", ".join(f"#{i}: {it[i]} it / {rounds[i]} rounds / {ev[i]} evals" for i in o))
    print(f"   time per round of the slowest: {ms * 1e3 / rounds.max():.2f} us")
