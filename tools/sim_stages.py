#!/usr/bin/env python3
"""List-scheduler replay of a launch (2048 resident wavefronts, workgroups start in index order) from the per-instance
evaluation counts at every outer-iteration boundary (tools/dump_outer_profile.py): what staging policies of the
resumable solve would buy.  usage: sim_stages.py <profile.npz> [slots]"""
import heapq
import sys

import numpy as np

d = np.load(sys.argv[1])
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
ev = d["evals"].astype(np.int64)            # cumulative after k = 1..10 outer iterations
B = ev.shape[0]
total = ev[:, -1]
ms_full = float(d["ms"][-1])


def replay(work, order=None):
    """makespan (in evaluations) of `work` items started in `order` on `slots` machines"""
    w = work if order is None else work[order]
    w = w[w > 0]
    if len(w) <= slots:
        return float(w.max()) if len(w) else 0.0
    h = list(w[:slots].astype(float))
    heapq.heapify(h)
    for x in w[slots:]:
        t = heapq.heappop(h)
        heapq.heappush(h, t + float(x))
    return max(h)


base = replay(total)
tau = ms_full / base                          # calibrate on the measured single launch
print(f"B={B} measured {ms_full:.1f} ms; work/capacity {total.sum() / slots * tau:.1f} ms; longest {total.max() * tau:.1f} ms; "
      f"perfect LPT {replay(total, np.argsort(-total)) * tau:.1f} ms")
inc = np.diff(np.concatenate([np.zeros((B, 1), np.int64), ev], axis=1), axis=1)   # evaluations spent in outer iteration k
for bounds in ([1], [2], [3], [2, 4], [2, 5], [2, 4, 6, 8], [1, 2, 3, 4, 5, 6, 7, 8, 9], [2, 4, 6, 7, 8, 9], [4, 7, 8, 9], [6, 8, 9], [7, 8, 9], [7], [8]):
    for rank in ("index", "evals"):
        t = 0.0
        prev = 0
        spent = np.zeros(B, np.int64)
        stages = bounds + [10]
        for s in stages:
            work = inc[:, prev:s].sum(axis=1)
            order = None
            if rank == "evals" and prev > 0:
                order = np.argsort(-spent, kind="stable")
            t += replay(work, order)
            spent += work
            prev = s
        print(f"  stages after outer {bounds} rank={rank}: {t * tau:.1f} ms (+ {len(bounds)} relaunches)")
