#!/usr/bin/env python3
"""configs[1], latency kernel: could ONE evaluation at the initial guess rank the instances as well as the pilot launch does?
List-scheduler replay (768 resident workgroups at W = 4; an instance's time ~ its exchange rounds) of a single launch
ordered by several keys, against the two-launch solve.   usage: exp_cfg1_prerank.py"""
import heapq, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dyobav_mpcnwta_warehouse_amd as nm
spec = dict(nm.scenarios.BENCH_CONFIGS["cfg1_b1024_n20_2x5"]); lay = spec.pop("layout"); B = spec.pop("B")
for fam in ("toward_robot", "passing"):
    P = nm.scenarios.make_batch_chunked(B, lay, ped_mode=fam, dtype=np.float32, **spec)
    def cfg_(**kw):
        cfg = nm.default_config_struct()
        cfg.N_hor, cfg.Nother, cfg.Nstcobs, cfg.Ndynobs = lay.N, lay.Nother, lay.Nstc, lay.Ndyn
        cfg.max_active_dynobs = spec["n_ped"] * spec["n_hyp"]; cfg.latency_waves = 4; cfg.staged = -1
        for k, v in kw.items(): setattr(cfg, k, v)
        return cfg
    with nm.Handle(cfg_()) as h:
        r = h.solve(P); r = h.solve(P); ms1 = h.last_kernel_ms()
        e0 = h.eval(P, np.zeros((B, 2 * lay.N), np.float32), np.zeros((B, 2 * lay.N), np.float32), np.full(B, 10.0, np.float32), grad=True)
    with nm.Handle(cfg_(max_outer_iterations=1)) as h:
        p1 = h.solve(P)
    with nm.Handle(cfg_(staged=0)) as h:
        h.solve(P); h.solve(P); ms2 = h.last_kernel_ms()
    rounds = r["info"][:, 6].astype(np.float64)
    rest = rounds - p1["info"][:, 6]
    def replay(work, order, slots=768):
        w = work[order]; w = w[w > 0]
        if len(w) <= slots: return w.max()
        hq = list(w[:slots]); heapq.heapify(hq)
        for x in w[slots:]: heapq.heappush(hq, heapq.heappop(hq) + x)
        return max(hq)
    tau = ms1 / replay(rounds, np.arange(B))
    gnorm = np.linalg.norm(e0["grad"], axis=1)
    keys = {"index order": np.arange(B), "perfect": np.argsort(-rounds), "||F2|| after the pilot (what the two-launch solve ranks by)": np.argsort(-p1["info"][:, 1]),
            "||F2(u0)||^2 from one evaluation": np.argsort(-e0["f2sq"]), "psi(u0)": np.argsort(-e0["psi"]), "||grad psi(u0)||": np.argsort(-gnorm)}
    print(f"{fam}: one launch in index order {ms1:.2f} ms (measured), two launches {ms2:.2f} ms (measured); longest instance {rounds.max() * tau:.2f} ms, work / 768 slots {rounds.sum() / 768 * tau:.2f} ms")
    for k, o in keys.items():
        print(f"   one launch ordered by {k:60s}: {replay(rounds, o) * tau:6.2f} ms (replay)")
    print(f"   two launches, replayed: pilot {replay(p1['info'][:, 6].astype(np.float64), np.arange(B)) * tau:.2f} + second launch ranked by the pilot's ||F2|| {replay(rest, np.argsort(-p1['info'][:, 1])) * tau:.2f} ms")
